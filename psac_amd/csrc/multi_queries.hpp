// multi_queries.hpp -- the members of MultiRun (multi.hpp) that work on a finished block-distributed result: all nearest smaller
// values (ansv<..., global_indexing>, ansv.hpp:2042-2051), the left-branching characters (par_rmq.hpp:334-481), the suffix-tree node
// table (suffix_tree.hpp:43-223, :440-499) and the distributed checker (check_suffix_array.hpp:207-267).  Declared in multi.hpp.
#pragma once
#include "multi.hpp"

namespace psacx {

// ---------------------------------------------------------------- all nearest smaller values over a block-distributed array
// ansv<T, left_type, right_type, global_indexing> (ansv.hpp:2042-2051; gansv_impl :1304-1740 keeps per-rank stacks
// and exchanges unmatched prefix minima).  Here every element first searches its own block (the tile kernel of
// ansv_wave.hpp); a search that leaves the block goes to the nearest further block whose all-gathered minimum
// qualifies and is answered from that block's edge.  furthest_eq = nearest <=, then the first strictly smaller value
// beyond it, then back to the first value <= (three searches, ansv_common.hpp:20-22).

template <typename T>
int MultiRun<T>::ansv_pyramid(int i, const T* block, uint64_t m, Pyramid<T>& Pm, DBuf<T>& mem, uint64_t* block_min) {
    psacx_ctx* c = ctx(i);
    Pm = Pyramid<T>();
    *block_min = ~0ull;
    if (m == 0) return PSACX_OK;
    uint64_t total = 0, len = m;
    while (len > 64) { len = (len + 63) / 64; total += (len + 63) & ~63ull; }
    MG_OP(g, c, mem.alloc(c, total + 64));
    Pm.lvl[0] = const_cast<T*>(block); Pm.len[0] = m; Pm.nlev = 1;
    len = m;
    uint64_t at = 0;
    OP_PROLOGUE(c);
    while (len > 64 && Pm.nlev < PYR_MAX) {
        len = (len + 63) / 64;
        Pm.lvl[Pm.nlev] = mem.p + at; Pm.len[Pm.nlev] = len; at += (len + 63) & ~63ull;
        hipLaunchKernelGGL((pyramid_level_kernel<T>), dim3(grid_for(c, len * 64, 256, 8)), dim3(256), 0, c->stream, Pm.lvl[Pm.nlev - 1],
                           Pm.len[Pm.nlev - 1], Pm.lvl[Pm.nlev], len);
        MG_HIP(g, hipGetLastError());
        Pm.nlev++;
    }
    unsigned long long* d = reinterpret_cast<unsigned long long*>(mem.p + at);
    hipLaunchKernelGGL((top_min_kernel<T>), dim3(1), dim3(256), 0, c->stream, Pm.lvl[Pm.nlev - 1], Pm.len[Pm.nlev - 1], d);
    MG_HIP(g, hipGetLastError());
    MG_HIP(g, hipMemcpyAsync(c->pinned + 32768, d, 8, hipMemcpyDeviceToHost, c->stream));
    MG_HIP(g, hipStreamSynchronize(c->stream));
    *block_min = *reinterpret_cast<uint64_t*>(c->pinned + 32768);
    return PSACX_OK;
}

// queries (start1 = start + 1, thr) of every local rank sent to rank cls[j] (< P; P = nowhere), answered there from
// that rank's block, answers back in query order.  idx / val: all ones / 0 where nothing was found or asked.
template <typename T>
int MultiRun<T>::ansv_ask(typename MultiRun<T>::AnsvState& A, const std::vector<const T*>& cls, const std::vector<const T*>& start1, const std::vector<const T*>& thr,
             const std::vector<uint64_t>& cnt, bool strict, bool left, std::vector<DBuf<T>>& idx, std::vector<DBuf<T>>& val) {
    std::vector<Rec<T>> ra(L), rb(L);
    std::vector<std::vector<uint64_t>> bounds(L), b2(L), rc, rc2;
    std::vector<std::vector<const T*>> in(L);
    std::vector<DBuf<T>> slot(L);
    PSACX_TRY(par([&](int i) -> int {
        psacx_ctx* c = ctx(i);
        MG_OP(g, c, slot[i].alloc(c, cnt[i]));
        MG_OP(g, c, psacx_op_iota(c, slot[i].p, cnt[i], 0));
        std::vector<uint64_t> bnd2;
        PSACX_TRY(route_by(i, cls[i], start1[i], thr[i], cnt[i], ra[i], bounds[i]));
        PSACX_TRY(route_by(i, cls[i], start1[i], slot[i].p, cnt[i], rb[i], bnd2));
        in[i] = {ra[i].k2.p, ra[i].v.p};
        return PSACX_OK;
    }));
    // class P ("nowhere") is the tail of the routed arrays: it is simply not sent (bounds[P] = its start)
    std::vector<std::vector<DBuf<T>>> q, got;
    PSACX_TRY(exchange<T>(2, in, bounds, q, rc));
    std::vector<DBuf<T>> ri(L), rv(L);
    PSACX_TRY(par([&](int i) -> int {
        psacx_ctx* c = ctx(i);
        const uint64_t qn = q[i][0].n;
        MG_OP(g, c, ri[i].alloc(c, qn)); MG_OP(g, c, rv[i].alloc(c, qn));
        OP_PROLOGUE(c);
        SIMPLE_LAUNCH(c, (nsv_from_enc_kernel<T>), qn, A.pyr[i], A.m[i], S[i].off, q[i][0].p, q[i][1].p, qn, strict ? 1 : 0, left ? 1 : 0, ri[i].p, rv[i].p);
        b2[i] = prefix_of(rc[i]);
        in[i] = {ri[i].p, rv[i].p};
        return PSACX_OK;
    }));
    PSACX_TRY(exchange<T>(2, in, b2, got, rc2));
    idx.clear(); idx.resize(L); val.clear(); val.resize(L);
    PSACX_TRY(par([&](int i) -> int {
        psacx_ctx* c = ctx(i);
        MG_OP(g, c, idx[i].alloc(c, cnt[i])); MG_OP(g, c, val[i].alloc(c, cnt[i]));
        OP_PROLOGUE(c);
        SIMPLE_LAUNCH(c, (fill_t_kernel<T>), cnt[i], idx[i].p, cnt[i], (T)~(T)0);
        SIMPLE_LAUNCH(c, (fill_t_kernel<T>), cnt[i], val[i].p, cnt[i], (T)0);
        const uint64_t back = got[i][0].n;            // answers come back for the queries that were sent, in routed order
        MG_OP(g, c, op_put(c, idx[i].p, rb[i].v.p, back, 0, got[i][0].p, 0));
        MG_OP(g, c, op_put(c, val[i].p, rb[i].v.p, back, 0, got[i][1].p, 0));
        return PSACX_OK;
    }));
    return PSACX_OK;
}

// For every query the nearest element strictly beyond start (start1 - 1; -1 and n allowed) with value < thr (strict) or
// <= thr, towards lower positions if left.  have_local: idx / val already hold the answers of the block that owns the
// start (the tile kernel's pass); otherwise that block is asked first.
template <typename T>
int MultiRun<T>::ansv_search(typename MultiRun<T>::AnsvState& A, const std::vector<const T*>& start1, const std::vector<const T*>& thr, const std::vector<uint64_t>& cnt,
                bool strict, bool left, bool have_local, std::vector<DBuf<T>>& idx, std::vector<DBuf<T>>& val) {
    const BlkDist bd = make_dist(n, (unsigned)P);
    std::vector<DBuf<T>> own(L);
    std::vector<const T*> cls(L);
    PSACX_TRY(par([&](int i) -> int {
        psacx_ctx* c = ctx(i);
        MG_OP(g, c, own[i].alloc(c, cnt[i]));
        OP_PROLOGUE(c);
        SIMPLE_LAUNCH(c, (ansv_owner_kernel<T>), cnt[i], start1[i], cnt[i], bd, own[i].p);
        cls[i] = own[i].p;
        return PSACX_OK;
    }));
    if (!have_local) PSACX_TRY(ansv_ask(A, cls, start1, thr, cnt, strict, left, idx, val));
    if (solo_) return PSACX_OK;
    RankMins rm, rs;
    for (int r = 0; r < 64; ++r) { rm.v[r] = r < P ? A.mins[r] : ~0ull; rs.v[r] = r < P ? sizes[r] : 0; }
    std::vector<DBuf<T>> target(L), edge(L);
    std::vector<const T*> tp(L), ep(L);
    PSACX_TRY(par([&](int i) -> int {
        psacx_ctx* c = ctx(i);
        MG_OP(g, c, target[i].alloc(c, cnt[i])); MG_OP(g, c, edge[i].alloc(c, cnt[i]));
        OP_PROLOGUE(c);
        SIMPLE_LAUNCH(c, (ansv_target_kernel<T>), cnt[i], own[i].p, thr[i], idx[i].p, cnt[i], rm, rs, P, strict ? 1 : 0, left ? 1 : 0, target[i].p);
        SIMPLE_LAUNCH(c, (fill_t_kernel<T>), cnt[i], edge[i].p, cnt[i], (T)(left ? n + 1 : 0));     // beyond the target's far edge
        tp[i] = target[i].p; ep[i] = edge[i].p;
        return PSACX_OK;
    }));
    std::vector<DBuf<T>> i2, v2;
    PSACX_TRY(ansv_ask(A, tp, ep, thr, cnt, strict, left, i2, v2));
    PSACX_TRY(par([&](int i) -> int {
        psacx_ctx* c = ctx(i);
        OP_PROLOGUE(c);
        SIMPLE_LAUNCH(c, (ansv_merge_kernel<T>), cnt[i], idx[i].p, val[i].p, i2[i].p, v2[i].p, target[i].p, cnt[i], P);
        return PSACX_OK;
    }));
    return PSACX_OK;
}

template <typename T>
int MultiRun<T>::ansv(const std::vector<const T*>& block, const std::vector<uint64_t>& m_local, int left_type, int right_type, uint64_t nonsv,
         const std::vector<uint64_t*>& out_left, const std::vector<uint64_t*>& out_right) {
    if (left_type < 0 || left_type > 2 || right_type < 0 || right_type > 2) return PSACX_EINVAL;
    S.resize(L);
    AnsvState A;
    A.block = block; A.m = m_local; A.pyr.resize(L); A.pyr_mem.resize(L);
    PSACX_TRY(par([&](int i) -> int {
        S[i].c = ctx(i); S[i].r = rank(i); S[i].m = m_local[i];
        MG_OP(g, S[i].c, ensure_pinned(S[i].c, 2 * sizeof(unsigned long long) * MAX_PASSES * RADIX + 65536 + 32768));
        return PSACX_OK;
    }));
    {
        std::vector<uint64_t> all;
        PSACX_TRY(gather1(m_local, all));
        sizes = all; offs = prefix_of(sizes); n = offs[P];
        for (int r = 0; r < P; ++r)
            if (sizes[r] != n / P + ((uint64_t)r < n % P ? 1 : 0)) { g->err = "The input must be equally block decomposed accross all MPI processes."; return PSACX_EINVAL; }
        for (int i = 0; i < L; ++i) S[i].off = offs[rank(i)];
        if (n == 0) return PSACX_EINVAL;
        if (diet) {
            // (a refinement step holds up to seventeen arrays of a slab's length at once -- its records, their new ids and the queries and
            //  answers of the range minima on both sides of an exchange -- beside the bucket ids and the list of unresolved positions:
            //  with 1/32 of a block per step that stays below three words per character, BASELINE.json configs[4])
            slab_cap = g->opt_slab ? g->opt_slab : slab_env_ ? slab_env_ : std::max<uint64_t>(sizes[0] / 32, 1u << 16);
            if (slab_cap < 64) slab_cap = 64;
            // (free blocks stay cached -- hipFree / hipMalloc of a 36 GB block cost about a second each -- and go back to the
            //  device only when an allocation does not fit: pool_alloc)
            for (int i = 0; i < L; ++i) ctx(i)->pool_cache_limit = 0;
            g->last_reduced = true;
        } else for (int i = 0; i < L; ++i) ctx(i)->pool_cache_limit = std::max<size_t>((size_t)S[i].m * sizeof(T) * 16, (size_t)64 << 20);   // free blocks kept for reuse: at most sixteen block-sized arrays (a flush is hipFree + hipMalloc of everything: seconds with eight ranks)
        if (sizeof(T) == 4 && n > 0xFFFFFFFDull) return PSACX_ERANGE;
    }
    std::vector<uint64_t> bm(L);
    for (int i = 0; i < L; ++i) PSACX_TRY(ansv_pyramid(i, block[i], m_local[i], A.pyr[i], A.pyr_mem[i], &bm[i]));
    PSACX_TRY(gather1(bm, A.mins));
    // every element's own position (plus one) as the start of its first search
    std::vector<DBuf<T>> here(L);
    std::vector<const T*> herep(L);
    PSACX_TRY(par([&](int i) -> int {
        MG_OP(g, ctx(i), here[i].alloc(ctx(i), m_local[i]));
        MG_OP(g, ctx(i), psacx_op_iota(ctx(i), here[i].p, m_local[i], S[i].off + 1));
        herep[i] = here[i].p;
        return PSACX_OK;
    }));
    for (int side = 0; side < 2; ++side) {
        const bool left = side == 0;
        const int typ = left ? left_type : right_type;
        const std::vector<uint64_t*>& out = left ? out_left : out_right;
        // first search inside the own block by the tile kernel (ansv_wave.hpp; a null array: that side is not computed)
        std::vector<DBuf<T>> idx(L), val(L);
        PSACX_TRY(par([&](int i) -> int {
            psacx_ctx* c = ctx(i);
            MG_OP(g, c, idx[i].alloc(c, m_local[i])); MG_OP(g, c, val[i].alloc(c, m_local[i]));
            if (!m_local[i]) return PSACX_OK;
            const int t1 = typ == 0 ? 0 : 1;                       // strict, or nearest <=
            MG_HIP(g, hipSetDevice(c->device));
            if (left) launch_ansv_tiles<T>(c, A.pyr[i], m_local[i], t1, 0, NSV_NONE, out[i], (uint64_t*)nullptr);
            else launch_ansv_tiles<T>(c, A.pyr[i], m_local[i], 0, t1, NSV_NONE, (uint64_t*)nullptr, out[i]);
            MG_HIP(g, hipGetLastError());
            OP_PROLOGUE(c);
            SIMPLE_LAUNCH(c, (ansv_local_to_idx_kernel<T>), m_local[i], out[i], block[i], m_local[i], S[i].off, idx[i].p, val[i].p);
            return PSACX_OK;
        }));
        PSACX_TRY(ansv_search(A, herep, block, m_local, typ == 0, left, true, idx, val));
        std::vector<DBuf<T>> far(L);
        if (typ == 2) {
            // s = first strictly smaller value beyond j (threshold: the value found at j), f = from s back towards i the first value <= it
            std::vector<DBuf<T>> st2(L), st3(L), si, sv, fv;
            std::vector<const T*> p2(L), p3(L), u(L);
            PSACX_TRY(par([&](int i) -> int {
                psacx_ctx* c = ctx(i);
                MG_OP(g, c, st2[i].alloc(c, m_local[i]));
                OP_PROLOGUE(c);
                SIMPLE_LAUNCH(c, (ansv_next_start_kernel<T>), m_local[i], idx[i].p, m_local[i], (T)(left ? n + 1 : 0), st2[i].p);
                p2[i] = st2[i].p; u[i] = val[i].p;
                return PSACX_OK;
            }));
            PSACX_TRY(ansv_search(A, p2, u, m_local, true, left, false, si, sv));
            PSACX_TRY(par([&](int i) -> int {
                psacx_ctx* c = ctx(i);
                MG_OP(g, c, st3[i].alloc(c, m_local[i]));
                OP_PROLOGUE(c);
                SIMPLE_LAUNCH(c, (ansv_next_start_kernel<T>), m_local[i], si[i].p, m_local[i], (T)(left ? 0 : n + 1), st3[i].p);
                p3[i] = st3[i].p;
                return PSACX_OK;
            }));
            PSACX_TRY(ansv_search(A, p3, u, m_local, false, !left, false, far, fv));
        }
        PSACX_TRY(par([&](int i) -> int {
            psacx_ctx* c = ctx(i);
            OP_PROLOGUE(c);
            SIMPLE_LAUNCH(c, (ansv_finish_kernel<T>), m_local[i], idx[i].p, typ == 2 ? (const T*)far[i].p : (const T*)idx[i].p, typ == 2 ? 1 : 0,
                          m_local[i], nonsv, out[i]);
            return PSACX_OK;
        }));
    }
    for (int i = 0; i < L; ++i) { MG_HIP(g, hipSetDevice(ctx(i)->device)); MG_HIP(g, hipStreamSynchronize(ctx(i)->stream)); }
    return PSACX_OK;
}

// Left-branching characters of a block-distributed SA / LCP (suffix_array.hpp:211-212; the reference fills local_Lc
// inside its LCP code, :1365-1383 and par_rmq.hpp:334-481; the result is by definition Lc[i] = S[SA[i-1] + LCP[i]],
// desa.hpp:262-264, '\0' past the end and at i = 0): the last SA entry of every block goes to its right neighbour, the
// text positions are fetched from their owners through the engine's bulk-RMA exchange (dist_take), piece by piece so
// that a block that is a large share of its device fits.
template <typename T>
int MultiRun<T>::left_chars(const std::vector<const uint8_t*>& text, const std::vector<uint64_t>& m_local, const std::vector<T*>& d_sa,
               const std::vector<T*>& d_lcp, const std::vector<uint8_t*>& d_lc) {
    want_lcp = true;
    S.resize(L);
    for (int i = 0; i < L; ++i) { (void)hipSetDevice(ctx(i)->device); pool_flush(ctx(i)); }
    PSACX_TRY(par([&](int i) -> int {
        S[i].c = ctx(i); S[i].r = rank(i); S[i].m = m_local[i]; S[i].text = text[i];
        S[i].SA = d_sa[i]; S[i].ISA = nullptr; S[i].LCP = d_lcp[i];
        MG_OP(g, S[i].c, ensure_pinned(S[i].c, 2 * sizeof(unsigned long long) * MAX_PASSES * RADIX + 65536 + 32768));
        return PSACX_OK;
    }));
    uint64_t chunks = 1;
    {
        std::vector<std::vector<uint64_t>> mine(L, std::vector<uint64_t>(2, 0));
        for (int i = 0; i < L; ++i) {
            int same = 0;
            for (int j = 0; j < L; ++j) same += ctx(j)->device == ctx(i)->device;
            size_t fr = 0, tot = 0;
            MG_HIP(g, hipSetDevice(ctx(i)->device));
            MG_HIP(g, hipMemGetInfo(&fr, &tot));
            // the widened text (1 word per character) stays; a piece wants about 12 words per entry
            const double avail = 0.8 * (double)fr / same - (double)m_local[i] * sizeof(T), need = 12.0 * (double)m_local[i] * sizeof(T);
            mine[i][0] = m_local[i];
            mine[i][1] = check_chunks_env_ ? check_chunks_env_ : need > avail ? (uint64_t)(need / std::max(avail, 1.0)) + 1 : 1;
        }
        std::vector<uint64_t> all;
        PSACX_TRY(gather(2, mine, all));
        sizes.assign(P, 0);
        for (int r = 0; r < P; ++r) { sizes[r] = all[(size_t)r * 2]; chunks = std::max(chunks, all[(size_t)r * 2 + 1]); }
        chunks = std::min<uint64_t>(chunks, 4096);
        offs = prefix_of(sizes); n = offs[P];
        for (int r = 0; r < P; ++r)
            if (sizes[r] != n / P + ((uint64_t)r < n % P ? 1 : 0)) { g->err = "The input string must be equally block decomposed accross all MPI processes."; return PSACX_EINVAL; }
        for (int i = 0; i < L; ++i) { S[i].off = offs[rank(i)]; ctx(i)->pool_cache_limit = 0; }
        if (n == 0) return PSACX_EINVAL;
    }
    std::vector<DBuf<T>> wide(L);
    PSACX_TRY(par([&](int i) -> int {
        psacx_ctx* c = ctx(i);
        MG_OP(g, c, wide[i].alloc(c, S[i].m));
        OP_PROLOGUE(c);
        SIMPLE_LAUNCH(c, (widen_text_kernel<T>), S[i].m, text[i], S[i].m, wide[i].p);
        return PSACX_OK;
    }));
    // SA of the entry before every block
    std::vector<psacx_boundary> edge;
    {
        std::vector<uint64_t> one(L);
        std::vector<const T*> a1(L), a2(L), a3(L);
        for (int i = 0; i < L; ++i) { one[i] = S[i].m ? 1 : 0; a1[i] = S[i].SA + (S[i].m ? S[i].m - 1 : 0); a2[i] = a1[i]; a3[i] = a1[i]; }
        PSACX_TRY(neighbours(a1, a2, a3, one, 1, edge));
    }
    std::vector<uint64_t> carry(L, 0);                       // SA of the last entry of the previous piece
    for (uint64_t q = 0; q < chunks; ++q) {
        std::vector<uint64_t> from(L), cnt(L);
        for (int i = 0; i < L; ++i) {
            from[i] = (uint64_t)(((unsigned __int128)S[i].m * q) / chunks);
            cnt[i] = (uint64_t)(((unsigned __int128)S[i].m * (q + 1)) / chunks) - from[i];
        }
        std::vector<DBuf<T>> qs(L), ch;
        std::vector<const T*> blk(L), gi(L);
        PSACX_TRY(par([&](int i) -> int {
            psacx_ctx* c = ctx(i);
            MG_OP(g, c, qs[i].alloc(c, cnt[i]));
            const int has_prev = from[i] ? 1 : edge[i].has_prev;
            const uint64_t prev = from[i] ? carry[i] : edge[i].prev[0];
            OP_PROLOGUE(c);
            SIMPLE_LAUNCH(c, (lc_queries_kernel<T>), cnt[i], S[i].SA + from[i], S[i].LCP + from[i], cnt[i], n, has_prev, (T)prev, qs[i].p);
            if (cnt[i]) { std::vector<uint64_t> o; PSACX_TRY(fetch(i, S[i].SA + from[i], {cnt[i] - 1}, o)); carry[i] = o[0]; }
            blk[i] = wide[i].p; gi[i] = qs[i].p;
            return PSACX_OK;
        }));
        PSACX_TRY(dist_take(blk, gi, cnt, ch));
        PSACX_TRY(par([&](int i) -> int {
            psacx_ctx* c = ctx(i);
            OP_PROLOGUE(c);
            SIMPLE_LAUNCH(c, (lc_narrow_kernel<T>), cnt[i], (const T*)ch[i].p, (const T*)qs[i].p, cnt[i], n, d_lc[i] + from[i]);
            MG_HIP(g, hipStreamSynchronize(c->stream));
            return PSACX_OK;
        }));
    }
    return PSACX_OK;
}

// Suffix-tree node table of a block-distributed SA / LCP (construct_suffix_tree on p ranks, suffix_tree.hpp:413-499): rank r
// receives the rows of the LCP indices of its block, nodes[i][(sigma + 1) columns], column c = the child reached through
// the character with alphabet code c (0 = end of text), leaves numbered n + i, 0 = none.  Parents from the distributed
// ANSV of LCP (suffix_tree.hpp:62), the LCP values at the parents and the edge characters S[SA[i] + lcp] through the bulk
// fetch (dist_take), the cells to the owners of the parents' rows like bulk_permute's (index, value) pairs.
// d_nodes == nullptr: only *sigma is computed (the size query of psacx_suffix_tree_*).
template <typename T>
int MultiRun<T>::suffix_tree(const std::vector<const uint8_t*>& text, const std::vector<uint64_t>& m_local, const std::vector<T*>& d_sa,
                const std::vector<T*>& d_lcp, const std::vector<unsigned long long*>* d_nodes, uint32_t* sigma) {
    want_lcp = true;
    // ---- alphabet over all blocks (alphabet.hpp:147-164: codes 1 .. sigma in byte order)
    CodeTable tab;
    {
        std::vector<std::vector<uint64_t>> mine(L, std::vector<uint64_t>(256, 0));
        PSACX_TRY(par([&](int i) -> int {
            psacx_ctx* c = ctx(i);
            MG_OP(g, c, ensure_pinned(c, 2 * sizeof(unsigned long long) * MAX_PASSES * RADIX + 65536 + 32768));
            DBuf<unsigned long long> h; MG_OP(g, c, h.alloc(c, 256));
            MG_HIP(g, hipSetDevice(c->device));
            MG_HIP(g, hipMemsetAsync(h.p, 0, 256 * 8, c->stream));
            if (m_local[i]) {
                hipLaunchKernelGGL((char_hist_kernel<256>), dim3(grid_for(c, m_local[i] / 16 + 1, 256, 8)), dim3(256), 0, c->stream, text[i], m_local[i], h.p);
                MG_HIP(g, hipGetLastError());
            }
            MG_HIP(g, hipMemcpyAsync(c->pinned + 32768, h.p, 256 * 8, hipMemcpyDeviceToHost, c->stream));
            MG_HIP(g, hipStreamSynchronize(c->stream));
            std::memcpy(mine[i].data(), c->pinned + 32768, 256 * 8);
            return PSACX_OK;
        }));
        std::vector<uint64_t> all;
        PSACX_TRY(gather(256, mine, all));
        uint16_t next = 1;
        for (int ch = 0; ch < 256; ++ch) {
            uint64_t tot = 0;
            for (int r = 0; r < P; ++r) tot += all[(size_t)r * 256 + ch];
            tab.c[ch] = tot ? next++ : (uint16_t)0;
        }
        *sigma = next - 1u;
    }
    if (!d_nodes) return PSACX_OK;
    const uint64_t row = (uint64_t)*sigma + 1;
    // ---- ANSV of LCP: left furthest_eq, right nearest_sm
    std::vector<DBuf<uint64_t>> ln(L), rn(L);
    {
        std::vector<const T*> blk(L); std::vector<uint64_t*> ol(L), orr(L);
        for (int i = 0; i < L; ++i) {
            MG_OP(g, ctx(i), ln[i].alloc(ctx(i), m_local[i])); MG_OP(g, ctx(i), rn[i].alloc(ctx(i), m_local[i]));
            blk[i] = d_lcp[i]; ol[i] = ln[i].p; orr[i] = rn[i].p;
        }
        PSACX_TRY(ansv(blk, m_local, 2, 0, NSV_NONE, ol, orr));          // (sets sizes / offs / n)
    }
    S.resize(L);
    for (int i = 0; i < L; ++i) {
        S[i].c = ctx(i); S[i].r = rank(i); S[i].m = m_local[i]; S[i].text = text[i]; S[i].SA = d_sa[i]; S[i].ISA = nullptr; S[i].LCP = d_lcp[i];
        S[i].off = offs[rank(i)];
    }
    // ---- LCP at the two parents; the first LCP entry of the next block
    std::vector<DBuf<T>> lcp_l, lcp_r;
    {
        std::vector<DBuf<T>> pl(L), pr(L);
        std::vector<const T*> blk(L), g1(L), g2(L);
        PSACX_TRY(par([&](int i) -> int {
            psacx_ctx* c = ctx(i);
            MG_OP(g, c, pl[i].alloc(c, S[i].m)); MG_OP(g, c, pr[i].alloc(c, S[i].m));
            OP_PROLOGUE(c);
            SIMPLE_LAUNCH(c, (st_nsv_positions_kernel<T>), S[i].m, (const uint64_t*)ln[i].p, S[i].m, pl[i].p);
            SIMPLE_LAUNCH(c, (st_nsv_positions_kernel<T>), S[i].m, (const uint64_t*)rn[i].p, S[i].m, pr[i].p);
            blk[i] = S[i].LCP; g1[i] = pl[i].p; g2[i] = pr[i].p;
            return PSACX_OK;
        }));
        PSACX_TRY(dist_take(blk, g1, m_local, lcp_l));
        PSACX_TRY(dist_take(blk, g2, m_local, lcp_r));
    }
    std::vector<psacx_boundary> edge;
    {
        std::vector<const T*> a1(L);
        for (int i = 0; i < L; ++i) a1[i] = S[i].LCP;
        PSACX_TRY(neighbours(a1, a1, a1, m_local, 1, edge));
    }
    // ---- parents and edge positions, edge characters
    std::vector<DBuf<T>> p1(L), p2(L);
    std::vector<DBuf<uint64_t>> q1(L), q2(L);
    PSACX_TRY(par([&](int i) -> int {
        psacx_ctx* c = ctx(i);
        MG_OP(g, c, p1[i].alloc(c, S[i].m)); MG_OP(g, c, p2[i].alloc(c, S[i].m)); MG_OP(g, c, q1[i].alloc(c, S[i].m)); MG_OP(g, c, q2[i].alloc(c, S[i].m));
        OP_PROLOGUE(c);
        SIMPLE_LAUNCH(c, (st_parents_kernel<T>), S[i].m, (const T*)S[i].LCP, (const T*)S[i].SA, S[i].m, S[i].off, n, (const uint64_t*)ln[i].p, (const uint64_t*)rn[i].p,
                      (const T*)lcp_l[i].p, (const T*)lcp_r[i].p, (int)edge[i].has_next, (T)edge[i].next[0], p1[i].p, q1[i].p, p2[i].p, q2[i].p);
        MG_HIP(g, hipStreamSynchronize(c->stream));
        return PSACX_OK;
    }));
    for (int i = 0; i < L; ++i) { ln[i].release(); rn[i].release(); lcp_l[i].release(); lcp_r[i].release(); }
    std::vector<DBuf<T>> wide(L);
    PSACX_TRY(par([&](int i) -> int {
        psacx_ctx* c = ctx(i);
        MG_OP(g, c, wide[i].alloc(c, S[i].m));
        MG_HIP(g, hipSetDevice(c->device));
        MG_HIP(g, hipMemsetAsync((*d_nodes)[i], 0, S[i].m * row * sizeof(unsigned long long), c->stream));
        OP_PROLOGUE(c);
        SIMPLE_LAUNCH(c, (widen_text_kernel<T>), S[i].m, text[i], S[i].m, wide[i].p);
        return PSACX_OK;
    }));
    for (int which = 0; which < 2; ++which) {
        std::vector<DBuf<uint64_t>>& q = which ? q2 : q1;
        std::vector<DBuf<T>>& par_ = which ? p2 : p1;
        std::vector<DBuf<T>> qs(L), ch, x(L), y(L);
        std::vector<const T*> blk(L), gi(L);
        PSACX_TRY(par([&](int i) -> int {
            psacx_ctx* c = ctx(i);
            MG_OP(g, c, qs[i].alloc(c, S[i].m));
            OP_PROLOGUE(c);
            SIMPLE_LAUNCH(c, (st_positions_kernel<T>), S[i].m, (const uint64_t*)q[i].p, S[i].m, n, qs[i].p);
            blk[i] = wide[i].p; gi[i] = qs[i].p;
            return PSACX_OK;
        }));
        PSACX_TRY(dist_take(blk, gi, m_local, ch));
        PSACX_TRY(par([&](int i) -> int {
            psacx_ctx* c = ctx(i);
            MG_OP(g, c, x[i].alloc(c, S[i].m)); MG_OP(g, c, y[i].alloc(c, S[i].m));
            OP_PROLOGUE(c);
            SIMPLE_LAUNCH(c, (st_payload_kernel<T>), S[i].m, (const uint64_t*)q[i].p, (const T*)ch[i].p, S[i].m, S[i].off, n, tab, which == 0 ? 1 : 0, x[i].p, y[i].p);
            return PSACX_OK;
        }));
        // the cells to the owners of their rows: the same stable partition by owner for both payload words
        std::vector<const T*> pos(L), xs(L), ys(L);
        std::vector<uint64_t> tot(L);
        std::vector<Rec<T>> r1(L), r2(L);
        std::vector<std::vector<DBuf<T>>> got1, got2;
        if (solo_) { for (int i = 0; i < L; ++i) { pos[i] = par_[i].p; xs[i] = x[i].p; ys[i] = y[i].p; tot[i] = S[i].m; } }
        else {
            std::vector<std::vector<uint64_t>> b1(L), b2(L), rc;
            std::vector<std::vector<const T*>> in1(L), in2(L);
            for (int i = 0; i < L; ++i) {
                PSACX_TRY(route(i, par_[i].p, x[i].p, S[i].m, r1[i], b1[i])); in1[i] = {r1[i].k2.p, r1[i].v.p};
                PSACX_TRY(route(i, par_[i].p, y[i].p, S[i].m, r2[i], b2[i])); in2[i] = {r2[i].v.p};
            }
            PSACX_TRY(exchange<T>(2, in1, b1, got1, rc));
            PSACX_TRY(exchange<T>(1, in2, b2, got2, rc));
            for (int i = 0; i < L; ++i) { pos[i] = got1[i][0].p; xs[i] = got1[i][1].p; ys[i] = got2[i][0].p; tot[i] = got1[i][0].n; }
        }
        PSACX_TRY(par([&](int i) -> int {
            psacx_ctx* c = ctx(i);
            OP_PROLOGUE(c);
            SIMPLE_LAUNCH(c, (st_put_kernel<T>), tot[i], (*d_nodes)[i], S[i].off, row, pos[i], xs[i], ys[i], tot[i], n);
            MG_HIP(g, hipStreamSynchronize(c->stream));
            return PSACX_OK;
        }));
    }
    return PSACX_OK;
}

// Distributed verification of block-distributed SA / ISA / LCP without gathering anything on one rank: what
// d_check_sa does (check_suffix_array.hpp:207-267: SA a permutation whose inverse is ISA, S[SA[i-1]] <= S[SA[i]],
// ties decided by the ranks of the suffixes one further) with the engine's own exchanges (bulk_rma for
// ISA[SA[i]], S[SA[i]], ISA[SA[i]+1]), plus the LCP array through its recurrence
//   LCP[i] = 0 | 1 | 1 + min(LCP[ISA[SA[i-1]+1]+1 .. ISA[SA[i]+1]])      (range minima: bulk_rmq_v2)
// which has the true LCP array as its only solution.  errors[0..3] as psacx_check_dev_*, summed over all ranks.
template <typename T>
int MultiRun<T>::check(const std::vector<const uint8_t*>& text, const std::vector<uint64_t>& m_local, const std::vector<T*>& d_sa,
          const std::vector<T*>& d_isa, const std::vector<T*>& d_lcp, bool with_lcp, uint64_t errors[4]) {
    want_lcp = with_lcp;
    S.resize(L);
    for (int i = 0; i < L; ++i) { (void)hipSetDevice(ctx(i)->device); pool_flush(ctx(i)); }     // the checker wants different sizes than the construction left cached
    PSACX_TRY(par([&](int i) -> int {
        S[i].c = ctx(i); S[i].r = rank(i); S[i].m = m_local[i]; S[i].text = text[i];
        S[i].SA = d_sa[i]; S[i].ISA = d_isa[i]; S[i].LCP = with_lcp ? d_lcp[i] : nullptr;
        MG_OP(g, S[i].c, ensure_pinned(S[i].c, 2 * sizeof(unsigned long long) * MAX_PASSES * RADIX + 65536 + 32768));
        return PSACX_OK;
    }));
    // The block is verified in `chunks` pieces of consecutive SA positions (every test is local to an entry and its
    // predecessor): one piece needs about 24 words per entry, so a block that large a share of the device is cut.
    uint64_t chunks = 1;
    {
        std::vector<std::vector<uint64_t>> mine(L, std::vector<uint64_t>(2, 0));
        for (int i = 0; i < L; ++i) {
            int same = 0;
            for (int j = 0; j < L; ++j) same += ctx(j)->device == ctx(i)->device;
            size_t fr = 0, tot = 0;
            MG_HIP(g, hipSetDevice(ctx(i)->device));
            MG_HIP(g, hipMemGetInfo(&fr, &tot));
            const double avail = 0.8 * (double)fr / same, need = 24.0 * (double)m_local[i] * sizeof(T);
            mine[i][0] = m_local[i];
            mine[i][1] = check_chunks_env_ ? check_chunks_env_ : need > avail ? (uint64_t)(need / std::max(avail, 1.0)) + 1 : 1;
        }
        std::vector<uint64_t> all;
        PSACX_TRY(gather(2, mine, all));
        sizes.assign(P, 0);
        for (int r = 0; r < P; ++r) { sizes[r] = all[(size_t)r * 2]; chunks = std::max(chunks, all[(size_t)r * 2 + 1]); }
        chunks = std::min<uint64_t>(chunks, 4096);
        offs = prefix_of(sizes); n = offs[P];
        for (int r = 0; r < P; ++r)
            if (sizes[r] != n / P + ((uint64_t)r < n % P ? 1 : 0)) { g->err = "The input string must be equally block decomposed accross all MPI processes."; return PSACX_EINVAL; }
        for (int i = 0; i < L; ++i) { S[i].off = offs[rank(i)]; ctx(i)->pool_cache_limit = 0; }
        if (n == 0) return PSACX_EINVAL;
    }
    // the text as index words, once (S[SA[i]] travels through the same exchanges as the indices)
    std::vector<DBuf<T>> wide(L);
    PSACX_TRY(par([&](int i) -> int {
        psacx_ctx* c = ctx(i);
        MG_OP(g, c, wide[i].alloc(c, S[i].m));
        OP_PROLOGUE(c);
        SIMPLE_LAUNCH(c, (widen_text_kernel<T>), S[i].m, text[i], S[i].m, wide[i].p);
        return PSACX_OK;
    }));
    // (SA, S[SA], ISA[SA + 1]) of a range of SA positions of every rank
    auto triple = [&](const std::vector<uint64_t>& from, const std::vector<uint64_t>& cnt, std::vector<DBuf<T>>* back, std::vector<DBuf<T>>& ch,
                      std::vector<DBuf<T>>& nx) -> int {
        std::vector<const T*> blk(L), gi(L);
        std::vector<DBuf<T>> q1(L);
        PSACX_TRY(par([&](int i) -> int {
            psacx_ctx* c = ctx(i);
            MG_OP(g, c, q1[i].alloc(c, cnt[i]));
            OP_PROLOGUE(c);
            SIMPLE_LAUNCH(c, (add_scalar_kernel<T>), cnt[i], S[i].SA + from[i], cnt[i], (uint64_t)1, n, q1[i].p);
            return PSACX_OK;
        }));
        for (int i = 0; i < L; ++i) { blk[i] = S[i].ISA; gi[i] = S[i].SA + from[i]; }
        if (back) PSACX_TRY(dist_take(blk, gi, cnt, *back));
        for (int i = 0; i < L; ++i) blk[i] = wide[i].p;
        PSACX_TRY(dist_take(blk, gi, cnt, ch));
        for (int i = 0; i < L; ++i) { blk[i] = S[i].ISA; gi[i] = q1[i].p; }
        PSACX_TRY(dist_take(blk, gi, cnt, nx));
        return PSACX_OK;
    };
    // the last entry of every block: the predecessor of the next non-empty block's first entry
    std::vector<psacx_boundary> edge;
    {
        std::vector<uint64_t> from(L), one(L);
        std::vector<DBuf<T>> ch, nx;
        for (int i = 0; i < L; ++i) { one[i] = S[i].m ? 1 : 0; from[i] = S[i].m ? S[i].m - 1 : 0; }
        PSACX_TRY(triple(from, one, nullptr, ch, nx));
        std::vector<const T*> a1(L), a2(L), a3(L);
        for (int i = 0; i < L; ++i) { a1[i] = S[i].SA + from[i]; a2[i] = ch[i].p; a3[i] = nx[i].p; }
        PSACX_TRY(neighbours(a1, a2, a3, one, 3, edge));
    }
    std::vector<std::vector<uint64_t>> mine(L, std::vector<uint64_t>(4, 0));
    std::vector<std::vector<uint64_t>> carry(L, std::vector<uint64_t>(3, 0));        // last entry of the previous piece
    for (uint64_t q = 0; q < chunks; ++q) {
        std::vector<uint64_t> from(L), cnt(L);
        for (int i = 0; i < L; ++i) {
            from[i] = (uint64_t)(((unsigned __int128)S[i].m * q) / chunks);
            cnt[i] = (uint64_t)(((unsigned __int128)S[i].m * (q + 1)) / chunks) - from[i];
        }
        std::vector<DBuf<T>> back, ch, nx, mins;
        PSACX_TRY(triple(from, cnt, &back, ch, nx));
        std::vector<psacx_boundary> bd(L);
        for (int i = 0; i < L; ++i) {
            std::memset(&bd[i], 0, sizeof(psacx_boundary));
            if (from[i] == 0) { bd[i].has_prev = edge[i].has_prev; for (int w = 0; w < 3; ++w) bd[i].prev[w] = edge[i].prev[w]; }
            else { bd[i].has_prev = 1; for (int w = 0; w < 3; ++w) bd[i].prev[w] = carry[i][w]; }
        }
        PSACX_TRY(par([&](int i) -> int {                     // this piece's last entry, for the next one
            if (!cnt[i]) return PSACX_OK;
            const T* arr[3] = {S[i].SA + from[i], ch[i].p, nx[i].p};
            for (int w = 0; w < 3; ++w) { std::vector<uint64_t> o; PSACX_TRY(fetch(i, arr[w], {cnt[i] - 1}, o)); carry[i][w] = o[0]; }
            return PSACX_OK;
        }));
        if (with_lcp) {
            std::vector<DBuf<T>> qlo(L), qhi(L);
            std::vector<const T*> lo(L), hi(L);
            PSACX_TRY(par([&](int i) -> int {
                psacx_ctx* c = ctx(i);
                MG_OP(g, c, qlo[i].alloc(c, cnt[i])); MG_OP(g, c, qhi[i].alloc(c, cnt[i]));
                OP_PROLOGUE(c);
                SIMPLE_LAUNCH(c, (check_queries_kernel<T>), cnt[i], S[i].SA + from[i], ch[i].p, nx[i].p, cnt[i], n, bd[i].has_prev, (T)bd[i].prev[0],
                              (T)bd[i].prev[1], (T)bd[i].prev[2], qlo[i].p, qhi[i].p);
                lo[i] = qlo[i].p; hi[i] = qhi[i].p;
                return PSACX_OK;
            }));
            PSACX_TRY(dist_range_min(lo, hi, cnt, mins));
        }
        PSACX_TRY(par([&](int i) -> int {
            psacx_ctx* c = ctx(i);
            DBuf<unsigned long long> e; MG_OP(g, c, e.alloc(c, 4));
            MG_HIP(g, hipSetDevice(c->device));
            MG_HIP(g, hipMemsetAsync(e.p, 0, 32, c->stream));
            OP_PROLOGUE(c);
            SIMPLE_LAUNCH(c, (check_verdict_kernel<T>), cnt[i], S[i].SA + from[i], back[i].p, ch[i].p, nx[i].p, with_lcp ? (const T*)(S[i].LCP + from[i]) : (const T*)nullptr,
                          with_lcp ? (const T*)mins[i].p : (const T*)nullptr, cnt[i], S[i].off + from[i], n, bd[i].has_prev, (T)bd[i].prev[0], (T)bd[i].prev[1],
                          (T)bd[i].prev[2], e.p);
            MG_HIP(g, hipMemcpyAsync(c->pinned + 32768, e.p, 32, hipMemcpyDeviceToHost, c->stream));
            MG_HIP(g, hipStreamSynchronize(c->stream));
            const uint64_t* h = reinterpret_cast<const uint64_t*>(c->pinned + 32768);
            for (int w = 0; w < 4; ++w) mine[i][w] += h[w];
            return PSACX_OK;
        }));
    }
    std::vector<uint64_t> all;
    PSACX_TRY(gather(4, mine, all));
    for (int q = 0; q < 4; ++q) { errors[q] = 0; for (int r = 0; r < P; ++r) errors[q] += all[(size_t)r * 4 + q]; }
    return PSACX_OK;
}

} // namespace psacx
