// multi_refine.hpp -- members of MultiRun (multi.hpp) that belong to the refinement rounds on p ranks: the bucket-local sort of a round's
// records and the range minima over the block-distributed LCP array.  Included by multi.hip after multi.hpp.
#pragma once
#include "multi.hpp"

namespace psacx {

// The sort of a refinement round: records (bucket id, rank h further, suffix) of the unresolved positions of every rank, in SA order.
// A bucket's records are neighbours before and after the sort, so only the buckets that reach over a rank boundary need their ranks
// to talk: those records (the first of a rank whose bucket started on a lower rank -- bucket id <= block offset --, the last of a rank
// whose bucket goes on on the next one) are sorted across the ranks (dist_sort on them alone: the per-rank counts are kept, and
// ascending bucket ids put every piece back in its place), all others by a local sort.  psac sorts its unresolved buckets the same
// way (suffix_array.hpp:1092-1157, the split buckets in two phases: stringset.hpp:323-375); a doubling round that sorts all records
// across the ranks (idxsort.hpp:23-83) moves three words per unresolved suffix over the links instead.
// The local part sorts two-word records: a bucket's records keep their places as a set, so the bucket ids (k1) stay where they are and
// only (rank h further, suffix) move, under the key (bucket's number in the run << bits2 | rank) -- as many digits as the run has
// buckets and the text has ranks, instead of both 64-bit words of a three-word record (a tandem repeat of 2^31 characters on
// 8 ranks: 40 key bits in 32-byte records instead of 66 in 48-byte ones).
template <typename T>
int MultiRun<T>::refine_sort(std::vector<Rec<T>>& rec, const std::vector<const T*>& plist, const std::vector<uint64_t>& counts, unsigned bits1, unsigned bits2) {
    if (global_refine_sort_env_) return dist_sort(rec, counts, bits1, bits2);
    // first / last bucket id of every rank (one rank: no bucket is shared, everything below is the local part)
    std::vector<std::vector<uint64_t>> mine(L, std::vector<uint64_t>(3, 0));
    PSACX_TRY(par([&](int i) -> int {
        if (!rec[i].cnt) return PSACX_OK;
        std::vector<uint64_t> o;
        PSACX_TRY(fetch(i, rec[i].k1.p, {0, rec[i].cnt - 1}, o));
        mine[i][0] = 1; mine[i][1] = o[0]; mine[i][2] = o[1];
        return PSACX_OK;
    }));
    std::vector<uint64_t> all;
    PSACX_TRY(gather(3, mine, all));
    std::vector<uint64_t> head(L, 0), tail(L, 0), ns(L, 0);
    PSACX_TRY(par([&](int i) -> int {
        const uint64_t cn = rec[i].cnt;
        if (!cn) return PSACX_OK;
        psacx_ctx* c = ctx(i);
        uint64_t next_first = 0;
        for (int r = rank(i) + 1; r < P; ++r) if (all[(size_t)r * 3]) { next_first = all[(size_t)r * 3 + 1]; break; }
        // records with id <= offset: below (offset + 1); records with the last id, if the next rank starts with it: from lower_bound(last id) on
        const bool goes_on = next_first != 0 && next_first == mine[i][2];
        MG_HIP(g, hipSetDevice(c->device));
        DBuf<uint64_t> d; MG_OP(g, c, d.alloc(c, 4));
        uint64_t* h = reinterpret_cast<uint64_t*>(c->pinned + 32768);
        h[0] = S[i].off + 1; h[1] = mine[i][2];
        MG_HIP(g, hipMemcpyAsync(d.p, h, 16, hipMemcpyHostToDevice, c->stream));
        hipLaunchKernelGGL((lower_bound_kernel<T>), dim3(1), dim3(64), 0, c->stream, (const T*)rec[i].k1.p, cn, (const uint64_t*)d.p, 2u, d.p + 2);
        MG_HIP(g, hipGetLastError());
        MG_HIP(g, hipMemcpyAsync(h, d.p + 2, 16, hipMemcpyDeviceToHost, c->stream));
        MG_HIP(g, hipStreamSynchronize(c->stream));
        head[i] = h[0];
        tail[i] = goes_on ? cn - h[1] : 0;
        if (head[i] + tail[i] >= cn) { head[i] = cn; tail[i] = 0; }          // (the whole rank lies in buckets shared with others)
        ns[i] = head[i] + tail[i];
        return PSACX_OK;
    }));
    std::vector<uint64_t> ns_all;
    PSACX_TRY(gather1(ns, ns_all));
    uint64_t any = 0;
    for (uint64_t x : ns_all) any += x;
    // the shared buckets' records across the ranks
    std::vector<Rec<T>> sh(L);
    if (any) {
        PSACX_TRY(par([&](int i) -> int {
            psacx_ctx* c = ctx(i);
            sh[i].cnt = ns[i];
            MG_OP(g, c, sh[i].k1.alloc(c, ns[i])); MG_OP(g, c, sh[i].k2.alloc(c, ns[i])); MG_OP(g, c, sh[i].v.alloc(c, ns[i]));
            MG_HIP(g, hipSetDevice(c->device));
            DBuf<T>* from[3] = {&rec[i].k1, &rec[i].k2, &rec[i].v};
            DBuf<T>* to[3] = {&sh[i].k1, &sh[i].k2, &sh[i].v};
            for (int q = 0; q < 3; ++q) {
                if (head[i]) MG_HIP(g, hipMemcpyAsync(to[q]->p, from[q]->p, head[i] * sizeof(T), hipMemcpyDeviceToDevice, c->stream));
                if (tail[i]) MG_HIP(g, hipMemcpyAsync(to[q]->p + head[i], from[q]->p + (rec[i].cnt - tail[i]), tail[i] * sizeof(T), hipMemcpyDeviceToDevice, c->stream));
            }
            return PSACX_OK;
        }));
        PSACX_TRY(dist_sort(sh, ns_all, bits1, bits2));
    }
    // everything else where it lies
    PSACX_TRY(par([&](int i) -> int {
        psacx_ctx* c = ctx(i);
        const uint64_t cn = rec[i].cnt, lo = head[i], len = cn - head[i] - tail[i];
        MG_HIP(g, hipSetDevice(c->device));
        const unsigned nb = bits_for(len > 2 ? (len - 1) >> 1 : 1);
        if (len >= 2 && nb + bits2 <= sizeof(T) * 8 && sizeof(T) == 8) {
            DBuf<T> ak, av;
            MG_OP(g, c, ak.alloc(c, len)); MG_OP(g, c, av.alloc(c, len));
            hipLaunchKernelGGL((refine_key_kernel<T>), dim3(grid_for(c, len, 256, 8)), dim3(256), 0, c->stream, plist[i] + lo, (const T*)(rec[i].k1.p + lo), rec[i].k2.p + lo, len, bits2);
            MG_HIP(g, hipGetLastError());
            int32_t where = 0;
            MG_OP(g, c, op_pair_sort<T>(c, rec[i].k2.p + lo, (T*)nullptr, rec[i].v.p + lo, ak.p, (T*)nullptr, av.p, len, nb + bits2, 0, &where));
            if (where) {
                MG_HIP(g, hipMemcpyAsync(rec[i].k2.p + lo, ak.p, len * sizeof(T), hipMemcpyDeviceToDevice, c->stream));
                MG_HIP(g, hipMemcpyAsync(rec[i].v.p + lo, av.p, len * sizeof(T), hipMemcpyDeviceToDevice, c->stream));
            }
            hipLaunchKernelGGL((mask_low_kernel<T>), dim3(grid_for(c, len, 256, 8)), dim3(256), 0, c->stream, rec[i].k2.p + lo, len, bits2);
            MG_HIP(g, hipGetLastError());
            MG_HIP(g, hipStreamSynchronize(c->stream));          // (the second record set goes back to the cache when this scope ends)
        } else if (len >= 2) {
            // (32-bit words, or keys too wide for one word: three arrays of the middle section's length only -- the records of the shared
            //  buckets at both ends are refilled from sh[] below -- and the sorted section copied back where the passes left it there)
            Rec<T> alt;
            PSACX_TRY(take3(i, alt, len));
            int32_t where = 0;
            MG_OP(g, c, op_pair_sort<T>(c, rec[i].k1.p + lo, rec[i].k2.p + lo, rec[i].v.p + lo, alt.k1.p, alt.k2.p, alt.v.p, len, bits1, bits2, &where));
            if (where) {
                MG_HIP(g, hipMemcpyAsync(rec[i].k1.p + lo, alt.k1.p, len * sizeof(T), hipMemcpyDeviceToDevice, c->stream));
                MG_HIP(g, hipMemcpyAsync(rec[i].k2.p + lo, alt.k2.p, len * sizeof(T), hipMemcpyDeviceToDevice, c->stream));
                MG_HIP(g, hipMemcpyAsync(rec[i].v.p + lo, alt.v.p, len * sizeof(T), hipMemcpyDeviceToDevice, c->stream));
                MG_HIP(g, hipStreamSynchronize(c->stream));
            }
            drop3(i, alt);
        }
        if (any && ns[i]) {
            if (sh[i].cnt != ns[i]) { mg_set_err(g, "refinement sort: the records of the shared buckets came back in other numbers"); return PSACX_EDEVICE; }
            DBuf<T>* to[3] = {&rec[i].k1, &rec[i].k2, &rec[i].v};
            DBuf<T>* from[3] = {&sh[i].k1, &sh[i].k2, &sh[i].v};
            for (int q = 0; q < 3; ++q) {
                if (head[i]) MG_HIP(g, hipMemcpyAsync(to[q]->p, from[q]->p, head[i] * sizeof(T), hipMemcpyDeviceToDevice, c->stream));
                if (tail[i]) MG_HIP(g, hipMemcpyAsync(to[q]->p + (cn - tail[i]), from[q]->p + head[i], tail[i] * sizeof(T), hipMemcpyDeviceToDevice, c->stream));
            }
            MG_HIP(g, hipStreamSynchronize(c->stream));          // (the small arrays go back to the cache when this scope ends)
        }
        return PSACX_OK;
    }));
    return PSACX_OK;
}

// min(LCP[lo .. hi)) over the block-distributed LCP array for every query (bulk_rmq_v2, par_rmq.hpp:199-332)
template <typename T>
int MultiRun<T>::dist_range_min(const std::vector<const T*>& lo, const std::vector<const T*>& hi, const std::vector<uint64_t>& cnt,
                   std::vector<DBuf<T>>& out) {
    out.clear(); out.resize(L);
    if (solo_) {
        psacx_ctx* c = ctx(0);
        MG_OP(g, c, out[0].alloc(c, cnt[0]));
        if (!cnt[0]) return PSACX_OK;
        if (cnt[0] >= S[0].m / 32) {          // (a whole round at once: a pyramid with the running minima of every level pays for itself)
            MG_OP(g, c, op_range_min<T>(c, S[0].LCP, S[0].m, lo[0], hi[0], cnt[0], S[0].off, out[0].p));
            return PSACX_OK;
        }
        Pyramid<T> Pm; uint64_t bmin = 0;
        PSACX_TRY(block_pyramid(0, Pm, &bmin));
        // many questions: the running minima of the groups of the upper levels beside the kept pyramid (a level then costs two loads
        // however short the range; the tables of level 0 -- two arrays of the block's length -- are not made: op_range_min makes them
        // for m / 32 questions and more, a slab has fewer)
        DBuf<T> aux;
        if (cnt[0] >= (1u << 16) && Pm.nlev > 2) {
            uint64_t tot = 0;
            for (int Lv = 1; Lv + 1 < Pm.nlev; ++Lv) tot += 2 * ((Pm.len[Lv] + 63) & ~63ull);
            MG_OP(g, c, aux.alloc(c, tot));
            OP_PROLOGUE(c);
            uint64_t at = 0;
            for (int Lv = 1; Lv + 1 < Pm.nlev; ++Lv) {
                T* pre = aux.p + at; at += (Pm.len[Lv] + 63) & ~63ull;
                T* suf = aux.p + at; at += (Pm.len[Lv] + 63) & ~63ull;
                hipLaunchKernelGGL((pyramid_aux_kernel<T>), dim3(grid_for(c, Pm.len[Lv], 256, 8)), dim3(256), 0, c->stream, Pm.lvl[Lv], Pm.len[Lv], pre, suf);
                MG_HIP(g, hipGetLastError());
                Pm.pre[Lv] = pre; Pm.suf[Lv] = suf;
            }
        }
        {
            OP_PROLOGUE(c);
            hipLaunchKernelGGL((range_min_kernel<T>), dim3(grid_for(c, cnt[0], 256, 16)), dim3(256), 0, c->stream, Pm, lo[0], hi[0], cnt[0], S[0].off, out[0].p);
            MG_HIP(g, hipGetLastError());
            if (aux.p) MG_HIP(g, hipStreamSynchronize(c->stream));          // (the tables go back to the cache when this scope ends)
        }
        return PSACX_OK;
    }
    // one min-pyramid of every rank's LCP block serves its block minimum and both batches of sub-queries
    std::vector<uint64_t> bm(L), mins;
    std::vector<Pyramid<T>> pyr(L);
    for (int i = 0; i < L; ++i) PSACX_TRY(block_pyramid(i, pyr[i], &bm[i]));
    PSACX_TRY(gather1(bm, mins));
    // a question is split into the part inside the rank of lo, the part inside the rank of hi - 1 (one half at a time: three arrays of
    // sub-questions alive, not six) and the whole ranks between, whose block minima every rank knows
    std::vector<std::vector<DBuf<T>>> parts(L);
    std::vector<std::vector<DBuf<T>>> answers(2);
    for (int half = 0; half < 2; ++half) {
        PSACX_TRY(par([&](int i) -> int {
            psacx_ctx* c = ctx(i);
            parts[i].clear(); parts[i].resize(6);
            for (int q = 0; q < 3; ++q) MG_OP(g, c, parts[i][3 * half + q].alloc(c, cnt[i]));
            OP_PROLOGUE(c);
            SIMPLE_LAUNCH(c, (rmq_split_half_kernel<T>), cnt[i], lo[i], hi[i], cnt[i], make_dist(n, (unsigned)P), half, parts[i][3 * half].p, parts[i][3 * half + 1].p,
                          parts[i][3 * half + 2].p);
            return PSACX_OK;
        }));
        std::vector<Rec<T>> ra(L), rb(L);
        std::vector<std::vector<uint64_t>> bounds(L), b2(L), rc, rc2;
        std::vector<std::vector<const T*>> in(L);
        PSACX_TRY(par([&](int i) -> int {
            psacx_ctx* c = ctx(i);
            const T* a = parts[i][3 * half + 1].p; const T* b = parts[i][3 * half + 2].p;
            // route by the owner of the sub-range's lower end: (a, b) and (a, slot) through the same stable pass
            DBuf<T> slot; MG_OP(g, c, slot.alloc(c, cnt[i]));
            MG_OP(g, c, psacx_op_iota(c, slot.p, cnt[i], 0));
            std::vector<uint64_t> bnd2;
            PSACX_TRY(route_by(i, parts[i][3 * half].p, a, b, cnt[i], ra[i], bounds[i]));
            PSACX_TRY(route_by(i, parts[i][3 * half].p, a, slot.p, cnt[i], rb[i], bnd2));
            rb[i].k2.release();                       // (only the slots of the second pass are read again)
            for (int q3 = 0; q3 < 3; ++q3) parts[i][3 * half + q3].release();      // this half's sub-queries are on their way
            in[i] = {ra[i].k2.p, ra[i].v.p};
            return PSACX_OK;
        }));
        std::vector<std::vector<DBuf<T>>> q, got;
        PSACX_TRY(exchange<T>(2, in, bounds, q, rc));
        ra.clear(); ra.resize(L);                     // (blocks go back to the rank's cache in stream order: engine.hpp pool)
        std::vector<DBuf<T>> res(L);
        PSACX_TRY(par([&](int i) -> int {
            psacx_ctx* c = ctx(i);
            MG_OP(g, c, res[i].alloc(c, q[i][0].n));
            OP_PROLOGUE(c);
            SIMPLE_LAUNCH(c, (range_min_kernel<T>), q[i][0].n, pyr[i], q[i][0].p, q[i][1].p, q[i][0].n, S[i].off, res[i].p);
            b2[i] = prefix_of(rc[i]);
            in[i] = {res[i].p};
            return PSACX_OK;
        }));
        PSACX_TRY(exchange<T>(1, in, b2, got, rc2));
        q.clear(); res.clear();
        answers[half].resize(L);
        PSACX_TRY(par([&](int i) -> int {
            psacx_ctx* c = ctx(i);
            MG_OP(g, c, answers[half][i].alloc(c, cnt[i]));
            MG_OP(g, c, op_put(c, answers[half][i].p, rb[i].v.p, cnt[i], 0, got[i][0].p, 0));
            return PSACX_OK;
        }));
    }
    RankMins rm;
    for (int r = 0; r < 64; ++r) rm.v[r] = r < P ? mins[r] : ~0ull;
    PSACX_TRY(par([&](int i) -> int {
        psacx_ctx* c = ctx(i);
        MG_OP(g, c, out[i].alloc(c, cnt[i]));
        OP_PROLOGUE(c);
        SIMPLE_LAUNCH(c, (rmq_combine_range_kernel<T>), cnt[i], answers[0][i].p, answers[1][i].p, lo[i], hi[i], cnt[i], make_dist(n, (unsigned)P), rm, out[i].p);
        return PSACX_OK;
    }));
    return PSACX_OK;
}

} // namespace psacx
