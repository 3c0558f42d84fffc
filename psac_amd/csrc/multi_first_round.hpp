// multi_first_round.hpp -- the first round of the multi-GPU engine in its two-word and one-word forms (members of MultiRun, declared
// in multi.hpp): the records are (word 1, suffix) or one 64-bit word, sorted on the leading bits of word 1 only; the suffixes that
// still tie fetch their full windows from the ranks that own their text (dist_windows) and are ordered by them (first_sort_ties,
// slab by slab in the reduced-memory layout).  Stands in for the first mxx::sort of idxsort.hpp:60-62 on (B1, B2, idx) tuples.
#pragma once
#include "multi.hpp"

namespace psacx {

// Both words of the packed 2k-character window of the suffixes gidx[i][0 .. cnt[i]) (global positions), computed by the
// ranks that own those positions from their text blocks + halos (tbuf: block + 2k characters) and sent back in query
// order: the remote form of window_word2() for the suffixes that tie on the leading bits of word 1.
template <typename T>
int MultiRun<T>::dist_windows(const std::vector<DBuf<uint8_t>>& tbuf, uint32_t two_k, const CodeTable& tab, const KeyShape& ks, const std::vector<const T*>& gidx,
                 const std::vector<uint64_t>& cnt, std::vector<DBuf<T>>& w1, std::vector<DBuf<T>>& w2) {
    w1.clear(); w1.resize(L); w2.clear(); w2.resize(L);
    auto answer = [&](int i, const T* q, uint64_t qn, T* o1, T* o2) -> int {
        psacx_ctx* c = ctx(i);
        OP_PROLOGUE(c);
        if (qn) {
            hipLaunchKernelGGL((window_at_kernel<T, 256>), dim3(grid_for(c, qn, 256, 16)), dim3(256), 0, c->stream, tbuf[i].p, S[i].m + two_k, S[i].off, q, qn,
                               tab, ks, o1, o2);
            PSACX_HIP(c, hipGetLastError());
        }
        return PSACX_OK;
    };
    if (solo_) {
        MG_OP(g, ctx(0), w1[0].alloc(ctx(0), cnt[0])); MG_OP(g, ctx(0), w2[0].alloc(ctx(0), cnt[0]));
        MG_OP(g, ctx(0), answer(0, gidx[0], cnt[0], w1[0].p, w2[0].p));
        return PSACX_OK;
    }
    std::vector<Rec<T>> routed(L);
    std::vector<std::vector<uint64_t>> bounds(L), rc, rc2;
    std::vector<std::vector<const T*>> in(L);
    PSACX_TRY(par([&](int i) -> int {
        psacx_ctx* c = ctx(i);
        DBuf<T> idx; MG_OP(g, c, idx.alloc(c, cnt[i]));
        MG_OP(g, c, psacx_op_iota(c, idx.p, cnt[i], 0));
        PSACX_TRY(route(i, gidx[i], idx.p, cnt[i], routed[i], bounds[i]));
        in[i] = {routed[i].k2.p};
        return PSACX_OK;
    }));
    std::vector<std::vector<DBuf<T>>> q, got;
    PSACX_TRY(exchange<T>(1, in, bounds, q, rc));
    std::vector<DBuf<T>> a1(L), a2(L);
    std::vector<std::vector<uint64_t>> back_bounds(L);
    PSACX_TRY(par([&](int i) -> int {
        psacx_ctx* c = ctx(i);
        MG_OP(g, c, a1[i].alloc(c, q[i][0].n)); MG_OP(g, c, a2[i].alloc(c, q[i][0].n));
        MG_OP(g, c, answer(i, q[i][0].p, q[i][0].n, a1[i].p, a2[i].p));
        back_bounds[i] = prefix_of(rc[i]);
        in[i] = {a1[i].p, a2[i].p};
        return PSACX_OK;
    }));
    PSACX_TRY(exchange<T>(2, in, back_bounds, got, rc2));
    PSACX_TRY(par([&](int i) -> int {
        psacx_ctx* c = ctx(i);
        MG_OP(g, c, w1[i].alloc(c, cnt[i])); MG_OP(g, c, w2[i].alloc(c, cnt[i]));
        MG_OP(g, c, op_put(c, w1[i].p, routed[i].v.p, cnt[i], 0, got[i][0].p, 0));      // undo the routing permutation
        MG_OP(g, c, op_put(c, w2[i].p, routed[i].v.p, cnt[i], 0, got[i][1].p, 0));
        return PSACX_OK;
    }));
    return PSACX_OK;
}

// The first sort in two-word form (what the one-GPU engine does, construct.hpp "two stages"): the records are (word 1,
// suffix) only.  When the leading `lead` = bits1 - lo1 bits of word 1 separate almost every suffix of the whole text,
//   1. the shuffle goes by those leading bits alone -- splitters are prefix values and equal prefixes never part, so a
//      group of suffixes that tie on them is whole on one rank -- and moves two words per record instead of three;
//   2. the local sort is a prefix sort of two-word records on the leading bits (lead / 8 passes of 4w bytes per record
//      instead of all digits of both words at 6w);
//   3. the few suffixes that still tie are compacted, the full window of each is fetched from the rank that owns its text
//      (dist_windows), the groups are ordered by it (in registers when every group is tiny, else by a radix sort of the
//      compacted records) and written back; word 2 exists for those records only, which is all rebucket_first_kernel reads.
// rec[i]: k1 and v filled, k2 allocated but unused until step 3.  Returns PSACX_RETRY_ before anything has moved when the
// samples say the text is repetitive (many equal prefixes) or the prefixes cannot balance the ranks: the caller then runs
// the three-word path.
template <typename T>
int MultiRun<T>::sort_first_two_word(std::vector<Rec<T>>& rec, const std::vector<uint64_t>& targets, unsigned bits1, unsigned bits2, unsigned lo1,
                        const std::vector<DBuf<uint8_t>>& tbuf, uint32_t two_k, const CodeTable& tab, const KeyShape& ks, bool trust, uint64_t spec_front) {
    ++sort_calls_;
    constexpr int SAMPLES = 8192;
    // Shuffle by key ranges (default with more than one rank): every destination's share of the prefix space is cut into QR
    // ranges, the block is partitioned once by (destination, range), range q of every destination travels in exchange q, and
    // the receiver sorts range q -- complete and final as soon as it has landed -- on its compute stream while ranges
    // q + 1 .. are still in flight on the second stream: the local sort runs under the shuffle (idxsort.hpp:58-62 sorts after
    // its Alltoallv has returned).  PSACX_MULTI_SHUFFLE_BY_POSITION=1: the earlier form (pieces of the block by position,
    // piece q + 1 partitioned while piece q travels, one local sort at the end).
    const bool by_range = !solo_;
    int QR = 1;
    if (by_range) { QR = std::max(1, std::min(4, 64 / P)); if (pieces_env_ > 0) QR = std::max(1, std::min(64 / P, pieces_env_)); }
    std::vector<uint64_t> spl;
    {
        std::vector<std::vector<uint64_t>> mine(L, std::vector<uint64_t>(1 + SAMPLES, 0));
        PSACX_TRY(par([&](int i) -> int {
            const uint64_t c = rec[i].cnt;
            std::vector<uint64_t> pos;
            for (int s = 0; s < SAMPLES && c; ++s) {
                const uint64_t lo = (uint64_t)(((unsigned __int128)c * s) / SAMPLES), hi = (uint64_t)(((unsigned __int128)c * (s + 1)) / SAMPLES);
                if (hi <= lo) continue;
                uint64_t z = ((uint64_t)rank(i) << 32 | (uint64_t)s) + 0x9E3779B97F4A7C15ull * (sort_calls_ + 1);      // splitmix64
                z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull; z = (z ^ (z >> 27)) * 0x94D049BB133111EBull; z ^= z >> 31;
                const uint64_t p = lo + z % (hi - lo);
                if (pos.empty() || pos.back() != p) pos.push_back(p);
            }
            std::vector<uint64_t> a;
            PSACX_TRY(fetch(i, rec[i].k1.p, pos, a));
            mine[i][0] = pos.size();
            for (size_t s = 0; s < pos.size(); ++s) mine[i][1 + s] = a[s] >> lo1;
            return PSACX_OK;
        }));
        std::vector<uint64_t> all, flat;
        PSACX_TRY(gather(1 + SAMPLES, mine, all));
        for (int r = 0; r < P; ++r) {
            const uint64_t* row = &all[(size_t)r * (1 + SAMPLES)];
            flat.insert(flat.end(), row + 1, row + 1 + row[0]);
        }
        std::sort(flat.begin(), flat.end());
        if (!trust && !flat.empty()) {
            // equal prefixes among a few thousand samples of a 2^lead space: a repetitive text, whose tie groups are long
            size_t dup = 0;
            for (size_t j = 1; j < flat.size(); ++j) dup += flat[j] == flat[j - 1];
            if (dup * 64 > flat.size()) return PSACX_RETRY_;
        }
        if (by_range) {
            // P * QR key ranges, QR consecutive ones per destination (equal splitters leave a range empty: the class numbers
            // must stay destination * QR + range)
            for (int cc = 1; cc < P * QR && !flat.empty(); ++cc) spl.push_back(flat[std::min(flat.size() - 1, flat.size() * cc / (size_t)(P * QR))]);
        } else {
            for (int d = 1; d < P && !flat.empty(); ++d) spl.push_back(flat[std::min(flat.size() - 1, flat.size() * d / P)]);
            spl.erase(std::unique(spl.begin(), spl.end()), spl.end());
        }
        if (!trust && !flat.empty() && P > 1) {
            // the share of the samples each destination would receive (destination = splitters <= prefix)
            std::vector<size_t> share(P, 0);
            for (uint64_t x : flat) share[std::min<size_t>((size_t)(std::upper_bound(spl.begin(), spl.end(), x) - spl.begin()) / (by_range ? QR : 1), P - 1)]++;
            for (int d = 0; d < P; ++d) if ((double)share[d] * P > 1.06 * (double)flat.size()) return PSACX_RETRY_;
        }
    }
    // The suffix a record stands for travels as a 32-bit entry while the text has at most 2^32 characters, else as a word.
    const bool v32 = sizeof(T) == 8 && n <= (1ull << 32);
    const size_t vb = v32 ? 4 : sizeof(T);
    // record j of local rank i stands for suffix: the spec short suffixes first on rank 0 (n - 1 - j), then the block in order
    auto payload_of = [&](int i, uint64_t a, uint64_t* spec_q, uint64_t* specn_q, uint64_t* voff_q) {
        const uint64_t front = rank(i) == 0 ? spec_front : 0;
        if (a == 0 && front) { *spec_q = front; *specn_q = n; *voff_q = 0; }          // (rank 0's block starts at position 0)
        else { *spec_q = 0; *specn_q = 0; *voff_q = S[i].off + a - front; }
    };
    bool sorted_already = false;
    if (by_range) {
        const int NC = P * QR;
        Splitters sp; std::memset(&sp, 0, sizeof(sp));
        sp.n = (uint32_t)spl.size();
        for (uint32_t s2 = 0; s2 < sp.n; ++s2) sp.k1[s2] = spl[s2];
        constexpr uint64_t SPAN = 256 * 32;
        std::vector<std::vector<uint64_t>> cnt_c(L, std::vector<uint64_t>((size_t)NC, 0));
        std::vector<DBuf<uint8_t>> cls(L);
        PSACX_TRY(par([&](int i) -> int {
            psacx_ctx* c = ctx(i);
            const uint64_t cn = rec[i].cnt;
            MG_OP(g, c, cls[i].alloc(c, cn + 16));
            DBuf<unsigned long long> d_counts; MG_OP(g, c, d_counts.alloc(c, 64));
            MG_HIP(g, hipSetDevice(c->device));
            MG_HIP(g, hipMemsetAsync(d_counts.p, 0, 64 * 8, c->stream));
            if (cn) {
                const uint64_t one = (cn + SPAN - 1) / SPAN * SPAN;           // the whole block as one "piece"
                hipLaunchKernelGGL((classify_prefix_kernel<T>), dim3((unsigned)(one / SPAN)), dim3(256), 0, c->stream, (const T*)rec[i].k1.p, cn, lo1, sp, cls[i].p, one, d_counts.p);
                MG_HIP(g, hipGetLastError());
            }
            MG_OP(g, c, ensure_pinned(c, 64 * 8 + 65536 + 32768));
            MG_HIP(g, hipMemcpyAsync(c->pinned + 32768, d_counts.p, 64 * 8, hipMemcpyDeviceToHost, c->stream));
            MG_HIP(g, hipStreamSynchronize(c->stream));
            const unsigned long long* h = reinterpret_cast<const unsigned long long*>(c->pinned + 32768);
            for (int cc = 0; cc < NC; ++cc) cnt_c[i][cc] = h[cc];
            return PSACX_OK;
        }));
        std::vector<uint64_t> table;                       // table[r * NC + destination * QR + range]
        PSACX_TRY(gather(NC, cnt_c, table));
        std::vector<Rec<T>> grp(L), rcv(L);
        std::vector<std::vector<uint64_t>> rbase(L), soff(L);          // start of range q in the receive arrays; start of class c in the partitioned block
        int rc_alloc = PSACX_OK;
        for (int i = 0; i < L && rc_alloc == PSACX_OK; ++i) {
            const int me = rank(i);
            rbase[i].assign(QR + 1, 0);
            for (int q = 0; q < QR; ++q) { uint64_t t = 0; for (int r = 0; r < P; ++r) t += table[(size_t)r * NC + me * QR + q]; rbase[i][q + 1] = rbase[i][q] + t; }
            soff[i] = prefix_of(cnt_c[i]);
            rc_alloc = take3(i, grp[i], rec[i].cnt, false);
        }
        PSACX_TRY(agree(rc_alloc));
        for (int i = 0; i < L && rc_alloc == PSACX_OK; ++i) rc_alloc = take3(i, rcv[i], rbase[i][QR], false);
        PSACX_TRY(agree(rc_alloc));
        // one stable partition of the block by class; the suffix a record stands for is made up on the way
        PSACX_TRY(par([&](int i) -> int {
            psacx_ctx* c = ctx(i);
            const uint64_t cn = rec[i].cnt;
            if (!cn) return PSACX_OK;
            SortScratch sc;
            auto layout = [&](Arena& ar) { sc.d_base = ar.take<unsigned long long>((size_t)RADIX); sc.desc_bytes = sort_desc_bytes(cn); sc.d_desc = ar.take<char>(sc.desc_bytes); };
            { Arena dry(nullptr); layout(dry); MG_OP(g, c, ensure_slab(c, dry.off + 4096)); }
            Arena ar(c->slab);
            layout(ar);
            uint64_t sq, snq, vq;
            payload_of(i, 0, &sq, &snq, &vq);
            MG_HIP(g, hipSetDevice(c->device));
            MG_OP(g, c, piece_partition<T>(c, sc.d_desc, sc.d_base, rec[i].k1.p, cls[i].p, cn, grp[i].k1.p, grp[i].v.p, v32, sq, snq, vq));
            return PSACX_OK;
        }));
        // the unpartitioned records are not needed any more: in the reduced-memory layout they sat in the rank's output arrays,
        // which now serve as the second record set of the range sorts
        for (int i = 0; i < L; ++i) { (void)hipSetDevice(ctx(i)->device); (void)hipStreamSynchronize(ctx(i)->stream); drop3(i, rec[i]); cls[i].release(); }
        std::vector<Rec<T>> alt(L);
        for (int i = 0; i < L && rc_alloc == PSACX_OK; ++i) rc_alloc = take3(i, alt[i], rbase[i][QR], false);
        PSACX_TRY(agree(rc_alloc));
        mark("    sort: classify + partition");
        // range q of every destination travels in exchanges 2 q (word 1) and 2 q + 1 (suffixes): all issued now, in order, on
        // the second streams.  The narrow suffix entries of range q land at the start of the range's own word-sized region,
        // so that the sort of an earlier range, which widens its entries in place, never touches a later range's input.
        std::vector<std::vector<hipEvent_t>> done(2 * QR, std::vector<hipEvent_t>(L, nullptr));
        auto drop_events = [&]() { for (auto& v : done) for (int i = 0; i < L; ++i) if (v[i]) { (void)hipSetDevice(ctx(i)->device); (void)hipEventDestroy(v[i]); v[i] = nullptr; } };
        int rc = PSACX_OK;
        for (int q = 0; q < 2 * QR && rc == PSACX_OK; ++q) for (int i = 0; i < L && rc == PSACX_OK; ++i)
            if (hipSetDevice(ctx(i)->device) != hipSuccess || hipEventCreateWithFlags(&done[q][i], hipEventDisableTiming) != hipSuccess) { mg_set_err(g, "two-word first sort: event creation failed"); rc = PSACX_EHIP; }
        const uint64_t wide = sizeof(T) / vb;                // narrow entries per word
        for (int q = 0; q < QR && rc == PSACX_OK; ++q) {
            for (int arr = 0; arr < 2 && rc == PSACX_OK; ++arr) {
                std::vector<std::vector<Msg>> sends(L), recvs(L);
                std::vector<std::vector<const void*>> in(L);
                std::vector<std::vector<void*>> out(L);
                for (int i = 0; i < L; ++i) {
                    const int me = rank(i);
                    for (int d = 0; d < P; ++d) sends[i].push_back(Msg{d, soff[i][(size_t)d * QR + q], cnt_c[i][(size_t)d * QR + q]});
                    uint64_t within = 0;
                    for (int r = 0; r < P; ++r) {
                        const uint64_t cn = table[(size_t)r * NC + me * QR + q];
                        recvs[i].push_back(Msg{r, (arr == 0 ? rbase[i][q] : rbase[i][q] * wide) + within, cn});
                        within += cn;
                    }
                    if (arr == 0) { in[i] = {grp[i].k1.p}; out[i] = {rcv[i].k1.p}; }
                    else { in[i] = {grp[i].v.p}; out[i] = {rcv[i].v.p}; }
                }
                rc = transfer(in, out, {arr == 0 ? sizeof(T) : vb}, sends, recvs, &done[2 * q + arr]);
            }
        }
        // the ranges, one after the other, as they arrive (a rank whose sort fails still waits for its messages and tells its peers:
        // every path below runs the waits, drops the events and agrees on the outcome)
        std::vector<std::vector<int32_t>> where(L, std::vector<int32_t>(QR, 0));
        for (int q = 0; q < QR && rc == PSACX_OK; ++q) {
            rc = (par([&](int i) -> int {
                psacx_ctx* c = ctx(i);
                MG_HIP(g, hipSetDevice(c->device));
                for (int s2 = 0; s2 < L; ++s2) { MG_HIP(g, hipStreamWaitEvent(c->stream, done[2 * q][s2], 0)); MG_HIP(g, hipStreamWaitEvent(c->stream, done[2 * q + 1][s2], 0)); }
                const uint64_t b0 = rbase[i][q], tq = rbase[i][q + 1] - b0;
                if (!tq) return PSACX_OK;
                MG_OP(g, c, op_pair_sort<T>(c, rcv[i].k1.p + b0, (T*)nullptr, rcv[i].v.p + b0, alt[i].k1.p + b0, (T*)nullptr, alt[i].v.p + b0, tq, bits1, 0, &where[i][q], lo1,
                                            false, 0, 0, v32));
                return PSACX_OK;
            }));
        }
        // everything has arrived (and, with ranks in one process, has been pulled) before the partitioned copies go away
        for (int i = 0; i < L; ++i) {
            (void)hipSetDevice(ctx(i)->device);
            for (int q = 0; q < 2 * QR; ++q) for (int s2 = 0; s2 < L; ++s2) if (done[q][s2]) (void)hipStreamWaitEvent(ctx(i)->stream, done[q][s2], 0);
        }
        for (int i = 0; i < L; ++i) { (void)hipSetDevice(ctx(i)->device); (void)hipStreamSynchronize(ctx(i)->stream); }
        drop_events();
        PSACX_TRY(agree(rc));
        // the sorted ranges into one record set (a sort's result lies in the set its last executed pass wrote)
        PSACX_TRY(par([&](int i) -> int {
            psacx_ctx* c = ctx(i);
            MG_HIP(g, hipSetDevice(c->device));
            int in_alt = 0;
            for (int q = 0; q < QR; ++q) in_alt += where[i][q] != 0;
            const bool to_alt = in_alt * 2 > QR;
            for (int q = 0; q < QR; ++q) {
                const uint64_t b0 = rbase[i][q], tq = rbase[i][q + 1] - b0;
                if (!tq || (where[i][q] != 0) == to_alt) continue;
                Rec<T>& from = to_alt ? rcv[i] : alt[i]; Rec<T>& to = to_alt ? alt[i] : rcv[i];
                MG_HIP(g, hipMemcpyAsync(to.k1.p + b0, from.k1.p + b0, tq * sizeof(T), hipMemcpyDeviceToDevice, c->stream));
                MG_HIP(g, hipMemcpyAsync(to.v.p + b0, from.v.p + b0, tq * sizeof(T), hipMemcpyDeviceToDevice, c->stream));
            }
            MG_HIP(g, hipStreamSynchronize(c->stream));
            drop3(i, grp[i]);
            if (to_alt) { drop3(i, rcv[i]); rec[i] = std::move(alt[i]); } else { drop3(i, alt[i]); rec[i] = std::move(rcv[i]); }
            rec[i].cnt = rbase[i][QR];
            return PSACX_OK;
        }));
        sorted_already = true;
        mark("    sort: shuffle by ranges + range sorts");
    }
    // prefix sort of (word 1, suffix) on the leading bits, then the ties
    PSACX_TRY(par([&](int i) -> int {
        psacx_ctx* c = ctx(i);
        if (rec[i].cnt >= 1 && !sorted_already) {
            Rec<T> alt;
            PSACX_TRY(take3(i, alt, rec[i].cnt, false));
            int32_t where = 0;
            if (solo_) {
                // the first pass makes up the payload (the suffix a record stands for), as on one GPU
                MG_OP(g, c, op_pair_sort<T>(c, rec[i].k1.p, (T*)nullptr, rec[i].v.p, alt.k1.p, (T*)nullptr, alt.v.p, rec[i].cnt, bits1, 0, &where, lo1, true, spec_front, n));
            } else MG_OP(g, c, op_pair_sort<T>(c, rec[i].k1.p, (T*)nullptr, rec[i].v.p, alt.k1.p, (T*)nullptr, alt.v.p, rec[i].cnt, bits1, 0, &where, lo1, false, 0, 0, v32));
            if (where) swap3(rec[i], alt);
            drop3(i, alt);
        }
        return PSACX_OK;
    }));
    return first_sort_ties(rec, targets, bits1, bits2, lo1, tbuf, two_k, tab, ks, false);
}

// Stage 2 of a first round that sorted (word 1, suffix) on the leading bits of word 1 only (rec[i]: k1, v sorted; word 1 may have lost the
// bits below the prefix: word1_gone): the suffixes that still tie are ordered by their full windows -- one rank with the text at hand: in
// place (tie_resolve_kernel); else compacted, their windows fetched from the ranks that own the text (dist_windows), ordered and written
// back -- and the records re-balanced to the block sizes.
template <typename T>
int MultiRun<T>::first_sort_ties(std::vector<Rec<T>>& rec, const std::vector<uint64_t>& targets, unsigned bits1, unsigned bits2, unsigned lo1,
                    const std::vector<DBuf<uint8_t>>& tbuf, uint32_t two_k, const CodeTable& tab, const KeyShape& ks, bool word1_gone) {
    std::vector<uint64_t> ties(L, 0);
    bool general_ties = !solo_;
    const bool solo_packed = word1_gone;
    PSACX_TRY(par([&](int i) -> int {
        psacx_ctx* c = ctx(i);
        PSACX_TRY(need_k2(i, rec[i]));
        if (solo_ && rec[i].cnt) {
            // one rank: the text is here, every tie group of at most 8 suffixes is ordered in place (tie_resolve_kernel, construct.hpp)
            constexpr int TB = 256, TI = sizeof(T) == 8 ? 32 : 16, TG = 8;
            DBuf<unsigned long long> big; MG_OP(g, c, big.alloc(c, 1));
            MG_HIP(g, hipSetDevice(c->device));
            MG_HIP(g, hipMemsetAsync(big.p, 0, 8, c->stream));
            const uint64_t nb = (rec[i].cnt + (uint64_t)TB * TI - 1) / ((uint64_t)TB * TI);
            hipLaunchKernelGGL((tie_resolve_kernel<T, TB, TI, TG>), dim3((unsigned)nb), dim3(TB), 0, c->stream, rec[i].k1.p, rec[i].v.p, rec[i].k2.p, rec[i].cnt, lo1,
                               (const uint8_t*)tbuf[i].p, S[i].m + two_k, tab, ks, big.p, solo_packed);
            MG_HIP(g, hipGetLastError());
            MG_HIP(g, hipMemcpyAsync(c->pinned + 32768, big.p, 8, hipMemcpyDeviceToHost, c->stream));
            MG_HIP(g, hipStreamSynchronize(c->stream));
            if (*reinterpret_cast<unsigned long long*>(c->pinned + 32768)) general_ties = true;
        }
        if (general_ties) MG_OP(g, c, op_compact_ties<T>(c, rec[i].k1.p, rec[i].v.p, rec[i].cnt, lo1, (T*)nullptr, (T*)nullptr, (T*)nullptr, &ties[i]));
        return PSACX_OK;
    }));
    mark("    sort: local prefix sort");
    if (!general_ties) { mark("    sort: ties"); return head_.empty() ? rebalance(rec, targets) : rebalance_in_place(rec, targets); }
    // Reduced-memory layout: the compacted ties, their windows and the second record set of their sort are eight arrays of as many
    // entries as there are ties -- on a repetitive text every suffix ties.  The records are then worked off in slabs of at most
    // `cap` records that end where a group of equal prefixes ends (groups are independent of each other; a group longer than a
    // slab is taken whole): the same steps on fewer records, every rank as many slabs as the one with the most.
    uint64_t cap = 0;
    if (diet && slab_cap) {
        std::vector<uint64_t> all;
        PSACX_TRY(gather1(ties, all));
        const uint64_t tcap = std::max<uint64_t>(slab_cap / 2, 64);
        for (uint64_t t : all) if (t > tcap) cap = tcap;
    }
    std::vector<uint64_t> at(L, 0), end(L, 0), tn(L, 0);
    for (;;) {
        if (!cap) for (int i = 0; i < L; ++i) { end[i] = rec[i].cnt; tn[i] = ties[i]; }
        else PSACX_TRY(par([&](int i) -> int {
            psacx_ctx* c = ctx(i);
            const uint64_t cnt = rec[i].cnt;
            end[i] = cnt; tn[i] = 0;
            if (at[i] >= cnt) return PSACX_OK;
            if (cnt - at[i] > cap) {
                DBuf<unsigned long long> cut; MG_OP(g, c, cut.alloc(c, 2));
                unsigned long long* h = reinterpret_cast<unsigned long long*>(c->pinned + 32768);
                MG_HIP(g, hipSetDevice(c->device));
                auto ask = [&](uint64_t lo, uint64_t hi) -> int {
                    h[0] = 0; h[1] = ~0ull;
                    MG_HIP(g, hipMemcpyAsync(cut.p, h, 16, hipMemcpyHostToDevice, c->stream));
                    hipLaunchKernelGGL((prefix_cut_kernel<T>), dim3(grid_for(c, hi - lo, 256, 8)), dim3(256), 0, c->stream, (const T*)rec[i].k1.p, lo, hi, lo1, cut.p, cut.p + 1);
                    MG_HIP(g, hipGetLastError());
                    MG_HIP(g, hipMemcpyAsync(h, cut.p, 16, hipMemcpyDeviceToHost, c->stream));
                    MG_HIP(g, hipStreamSynchronize(c->stream));
                    return PSACX_OK;
                };
                PSACX_TRY(ask(at[i] + 1, at[i] + cap + 1));              // the last group start inside the slab ...
                if (h[0]) end[i] = h[0];
                else {                                                   // ... or, a group longer than the slab, the end of that group
                    PSACX_TRY(ask(at[i] + cap + 1, cnt));
                    if (h[1] != ~0ull) end[i] = h[1];
                }
            }
            MG_OP(g, c, op_compact_ties<T>(c, rec[i].k1.p + at[i], rec[i].v.p + at[i], end[i] - at[i], lo1, (T*)nullptr, (T*)nullptr, (T*)nullptr, &tn[i]));
            return PSACX_OK;
        }));
        std::vector<DBuf<T>> tpos(L), tk1(L), tv(L), w1, w2;
        PSACX_TRY(par([&](int i) -> int {
            psacx_ctx* c = ctx(i);
            MG_OP(g, c, tpos[i].alloc(c, tn[i])); MG_OP(g, c, tk1[i].alloc(c, tn[i])); MG_OP(g, c, tv[i].alloc(c, tn[i]));
            if (tn[i]) { uint64_t chk = 0; MG_OP(g, c, op_compact_ties<T>(c, rec[i].k1.p + at[i], rec[i].v.p + at[i], end[i] - at[i], lo1, tpos[i].p, tk1[i].p, tv[i].p, &chk, /*counted=*/true)); }
            return PSACX_OK;
        }));
        {
            std::vector<const T*> q(L);
            for (int i = 0; i < L; ++i) q[i] = tv[i].p;
            PSACX_TRY(dist_windows(tbuf, two_k, tab, ks, q, tn, w1, w2));
        }
        PSACX_TRY(par([&](int i) -> int {
            psacx_ctx* c = ctx(i);
            const uint64_t n_t = tn[i];
            if (!n_t) return PSACX_OK;
            MG_HIP(g, hipSetDevice(c->device));
            // every group is at most TG long: ordered in registers (tie_resolve_kernel reading both words from the arrays)
            constexpr int TB_ = 256, TI_ = 16, TG_ = 8;
            DBuf<unsigned long long> big; MG_OP(g, c, big.alloc(c, 1));
            MG_HIP(g, hipMemsetAsync(big.p, 0, 8, c->stream));
            const uint64_t nb = (n_t + (uint64_t)TB_ * TI_ - 1) / ((uint64_t)TB_ * TI_);
            hipLaunchKernelGGL((tie_resolve_kernel<T, TB_, TI_, TG_, true>), dim3((unsigned)nb), dim3(TB_), 0, c->stream, w1[i].p, tv[i].p, w2[i].p, n_t, lo1,
                               (const uint8_t*)nullptr, (uint64_t)0, tab, ks, big.p);
            MG_HIP(g, hipGetLastError());
            MG_HIP(g, hipMemcpyAsync(c->pinned + 32768, big.p, 8, hipMemcpyDeviceToHost, c->stream));
            MG_HIP(g, hipStreamSynchronize(c->stream));
            const T *s1 = w1[i].p, *s2 = w2[i].p, *sv = tv[i].p;
            DBuf<T> b1, b2, bv;
            if (*reinterpret_cast<unsigned long long*>(c->pinned + 32768)) {
                // some group is long (repetitive text): a stable sort of all tied records by the full window; the groups come in
                // ascending order of their prefix, so the sorted records go back to the same positions in order
                tk1[i].release();                                    // (word 1 of the ties came back with the windows)
                MG_OP(g, c, b1.alloc(c, n_t)); MG_OP(g, c, b2.alloc(c, n_t)); MG_OP(g, c, bv.alloc(c, n_t));
                int32_t where = 0;
                MG_OP(g, c, op_pair_sort<T>(c, w1[i].p, w2[i].p, tv[i].p, b1.p, b2.p, bv.p, n_t, bits1, bits2, &where));
                if (where) { s1 = b1.p; s2 = b2.p; sv = bv.p; }
            }
            hipLaunchKernelGGL((scatter_prefix_ties_kernel<T>), dim3(grid_for(c, n_t, 256, 16)), dim3(256), 0, c->stream, (const T*)tpos[i].p, n_t, s1, s2, sv,
                               rec[i].k1.p + at[i], rec[i].k2.p + at[i], rec[i].v.p + at[i]);
            MG_HIP(g, hipGetLastError());
            MG_HIP(g, hipStreamSynchronize(c->stream));          // (the compacted arrays go back to the cache when this scope ends)
            return PSACX_OK;
        }));
        if (!cap) break;
        std::vector<uint64_t> left(L), left_all;
        for (int i = 0; i < L; ++i) { at[i] = end[i]; left[i] = rec[i].cnt - at[i]; }
        PSACX_TRY(gather1(left, left_all));
        bool more = false;
        for (uint64_t x : left_all) more |= x != 0;
        if (!more) break;
        ++g->last_tie_slabs;
    }
    mark("    sort: ties");
    return head_.empty() ? rebalance(rec, targets) : rebalance_in_place(rec, targets);
}

// The first sort in ONE-word records (the one-GPU engine's prefix_sort_1w, engine.hpp, spread over the ranks).  A record is
// (prefix of word 1 without its top digit) << sfield | suffix; the top digit is known from the record's place:
//   1. every rank counts the top digits of its block straight from the text (top_digit_hist_kernel); one all-gather of the 256 counts
//      gives every rank the exact size of every bucket on every rank -- no samples, no splitters;
//   2. the 256 buckets are dealt to the ranks in order, whole, so that every rank's share is as close to its block as whole buckets
//      allow (equal prefixes never part; the text's own distribution decides the balance: a text whose buckets cannot be dealt
//      within the slack of the record arrays takes the two-word path with its sampled splitters);
//   3. the pass on the top digit computes word 1 in registers and writes the one-word records bucket by bucket
//      (key_scatter1w_kernel): 1 byte read + 8 written per record, nothing else is ever written on the sender;
//   4. the buckets travel in QR groups per destination, each bucket's pieces from all senders landing back to back; a group is
//      complete when it has landed and its LSD passes (8 + 8 bytes per record and pass, radix_scatter1w_kernel) run on the compute
//      stream while the later groups are still in flight; the last pass writes word 1 and the suffixes as words.
// The suffixes shorter than 2k (the last 2k - 1 positions of the text) are made on the host -- every rank knows the tail of the text
// from the gather -- and placed at the head of their buckets, where the stable passes keep them in front of equal prefixes.
// Needs 64-bit words and n <= 2^34 (the payload field takes bits_for(n - 1) bits, the prefix the rest + 8: fewer than 1/16 of the suffixes
// of a random text tie).  Returns PSACX_RETRY_ before anything has moved.  *lo1_out: bits of word 1 below the sorted prefix.
template <typename T>
int MultiRun<T>::sort_first_one_word(std::vector<Rec<T>>& rec, const std::vector<uint64_t>& targets, unsigned bits1, unsigned bits2,
                        const std::vector<DBuf<uint8_t>>& tbuf, uint32_t two_k, const CodeTable& tab, const KeyShape& ks, bool trust, uint64_t spec,
                        unsigned* lo1_out) {
    if constexpr (sizeof(T) != 8) { return PSACX_RETRY_; }
    else {
    constexpr int BLOCK = 512, ITEMS = 8, TILE0 = BLOCK * ITEMS, TILE = BLOCK * PSACX_1W_ITEMS;
    constexpr int TAILB = 128;                                   // bytes of every block's end that travel with the counts (2k <= 128)
    const unsigned nbits = bits_for(n - 1);
    if (bits1 < 24 || nbits > 40) return PSACX_RETRY_;
    // prefix bits that stay in the word: what the one-GPU rule asks for (bits_for(n - 1) + 3 leading bits, whole digits) as far as the word has room
    const unsigned want_lead = (nbits + 3 + RADIX_BITS - 1) / RADIX_BITS * RADIX_BITS;
    const unsigned low = std::min(std::min(64u - nbits, bits1 - (unsigned)RADIX_BITS), want_lead - (unsigned)RADIX_BITS);
    const unsigned lead = low + RADIX_BITS, sfield = 64 - low, lo1 = bits1 - lead;
    // Reduced-memory layout: a text that repeats itself, or one so long that few prefix bits fit beside the suffix (beyond 2^34 characters),
    // stays in one-word records -- its many ties are ordered slab by slab (first_sort_ties), while the three-word records of the
    // other forms would not fit the device at all (8.25 words per character against 3)
    const bool ties_ok = trust || diet;
    if (lead < nbits + 3 && !ties_ok) return PSACX_RETRY_;      // (too many suffixes would tie on the prefix)
    if (lead < nbits + 1) return PSACX_RETRY_;
    uint64_t min_m = sizes[0];
    for (int r = 1; r < P; ++r) min_m = std::min(min_m, sizes[r]);
    if (min_m < (uint64_t)TAILB || min_m < 2ull * two_k) return PSACX_RETRY_;
    ++sort_calls_;
    KeyShape ks0 = ks; ks0.spec = solo_ ? spec : 0;               // (one rank without the wire: the short suffixes are records of the kernel, as on one GPU)
    // ---- 1. top digits of every block
    std::vector<uint64_t> nrec(L), short_n(L);
    std::vector<std::vector<uint64_t>> mine(L, std::vector<uint64_t>(RADIX + 2 + TAILB / 8, 0));
    struct Scr { unsigned long long* base0; char* desc; unsigned* tile_hist0; unsigned long long* slab_tot0; uint64_t ntiles; unsigned slab0; size_t desc_bytes; };
    std::vector<Scr> scr(L);
    PSACX_TRY(par([&](int i) -> int {
        psacx_ctx* c = ctx(i);
        const uint64_t m = S[i].m, end = S[i].off + m, first_short = n - spec;
        short_n[i] = solo_ ? 0 : std::min<uint64_t>(m, end > first_short ? end - first_short : 0);
        nrec[i] = m - short_n[i];
        Scr& q = scr[i];
        q.ntiles = (nrec[i] + TILE0 - 1) / TILE0;
        q.slab0 = slab_tiles_for(q.ntiles);
        // the scratch of the bucket passes on the receiving side lives in the same slab: sized now for the largest share a rank may accept
        const uint64_t cap_rec = m + m / 8 + 256 + (uint64_t)TILE;
        const uint64_t vt_ub = (cap_rec + TILE - 1) / TILE + (uint64_t)RADIX * 64 + 64;
        const size_t need_b = 256 + (((size_t)vt_ub * RADIX * sizeof(unsigned) + 255) & ~(size_t)255) + (((size_t)(vt_ub / 16 + RADIX) * RADIX * 8 + 255) & ~(size_t)255) +
                              (size_t)RADIX * RADIX * 8 + 2 * (RADIX + 1) * 8 + 64 + (size_t)(vt_ub / 16 + RADIX) * sizeof(SlabInfo) + 4096;
        const uint64_t stride = std::max<uint64_t>(64, m >> 20), samples = m / stride;
        uint64_t slots = 1; while (slots < 4 * samples) slots <<= 1;
        const size_t need_a = 256 + std::max<size_t>((((size_t)q.ntiles * RADIX * sizeof(unsigned) + 255) & ~(size_t)255) + (q.ntiles / q.slab0 + 2) * RADIX * 8, slots * 8) + 4096;
        q.desc_bytes = std::max(need_a, need_b);
        MG_OP(g, c, ensure_slab(c, q.desc_bytes + (size_t)RADIX * 8 + 8192));
        MG_OP(g, c, ensure_pinned(c, 2 * sizeof(unsigned long long) * MAX_PASSES * RADIX + 65536 + 32768));
        Arena ar(c->slab);
        q.base0 = ar.take<unsigned long long>((size_t)RADIX);
        q.desc = ar.take<char>(q.desc_bytes);
        q.tile_hist0 = reinterpret_cast<unsigned*>(q.desc + 256);
        q.slab_tot0 = reinterpret_cast<unsigned long long*>(q.desc + 256 + (((size_t)q.ntiles * RADIX * sizeof(unsigned) + 255) & ~(size_t)255));
        MG_HIP(g, hipSetDevice(c->device));
        unsigned long long* h = reinterpret_cast<unsigned long long*>(c->pinned + 32768);
        h[RADIX] = 0; h[RADIX + 1] = 0;
        if (samples >= 1024 && !ties_ok) {
            // does the block repeat itself massively?  (prefix_dup_probe_kernel, sa_kernels.hpp: such a text keeps the two-word path)
            unsigned long long* table = reinterpret_cast<unsigned long long*>(q.desc + 256);
            unsigned long long* d_dups = reinterpret_cast<unsigned long long*>(q.desc + 128);
            MG_HIP(g, hipMemsetAsync(q.desc, 0, 256 + slots * 8, c->stream));
            hipLaunchKernelGGL((prefix_dup_probe_kernel<uint64_t>), dim3((unsigned)((samples + 255) / 256)), dim3(256), 0, c->stream, (const uint8_t*)tbuf[i].p, m + two_k, tab, ks0, lo1,
                               stride, samples, table, slots, d_dups);
            MG_HIP(g, hipGetLastError());
            MG_HIP(g, hipMemcpyAsync(h + RADIX, d_dups, 8, hipMemcpyDeviceToHost, c->stream));
            MG_HIP(g, hipStreamSynchronize(c->stream));
            h[RADIX + 1] = samples;
        }
        if (q.ntiles) {
            hipLaunchKernelGGL((top_digit_hist_kernel<uint64_t, BLOCK, ITEMS>), dim3((unsigned)q.ntiles), dim3(BLOCK), 0, c->stream, (const uint8_t*)tbuf[i].p, solo_ ? m : nrec[i],
                               m + two_k, tab, ks0, q.tile_hist0);
            const uint64_t nslabs0 = (q.ntiles + q.slab0 - 1) / q.slab0;
            hipLaunchKernelGGL(radix_slab_scan_kernel<0>, dim3((unsigned)nslabs0), dim3(RADIX), 0, c->stream, q.tile_hist0, q.ntiles, q.slab_tot0, q.slab0);
            hipLaunchKernelGGL(radix_top_scan_kernel<0>, dim3(1), dim3(RADIX), 0, c->stream, q.slab_tot0, nslabs0, q.base0);
            MG_HIP(g, hipGetLastError());
            MG_HIP(g, hipMemcpyAsync(h, q.base0, RADIX * 8, hipMemcpyDeviceToHost, c->stream));
        } else std::memset(h, 0, RADIX * 8);
        MG_HIP(g, hipMemcpyAsync(h + RADIX + 2, tbuf[i].p + m - TAILB, TAILB, hipMemcpyDeviceToHost, c->stream));
        MG_HIP(g, hipStreamSynchronize(c->stream));
        const uint64_t total = solo_ ? m : nrec[i];
        for (int d = 0; d < RADIX; ++d) mine[i][d] = (d + 1 < RADIX ? h[d + 1] : total) - h[d];       // bucket sizes (the starts are their prefix sums)
        for (int w = RADIX; w < RADIX + 2 + TAILB / 8; ++w) mine[i][w] = h[w];
        return PSACX_OK;
    }));
    std::vector<uint64_t> table;                                 // table[r * W + b]
    const int W = RADIX + 2 + TAILB / 8;
    PSACX_TRY(gather(W, mine, table));
    // ---- 2. the short suffixes (host), the buckets' sizes, their owners
    {
        uint64_t dups = 0, smp = 0;
        for (int r = 0; r < P; ++r) { dups += table[(size_t)r * W + RADIX]; smp += table[(size_t)r * W + RADIX + 1]; }
        if (!ties_ok && smp && dups * 8 > smp) return PSACX_RETRY_;
    }
    std::vector<std::vector<uint64_t>> short_words(RADIX);
    if (!solo_ && spec) {
        const uint8_t* tail = reinterpret_cast<const uint8_t*>(&table[(size_t)(P - 1) * W + RADIX + 2]);       // text[n - TAILB .. n)
        for (uint64_t j = 0; j < spec; ++j) {                    // suffix n - 1 - j, j + 1 characters long: shortest first
            const uint64_t pos = n - 1 - j;
            uint64_t w1 = 0;
            for (unsigned t = 0; t < ks.c1; ++t) {
                const uint64_t code = pos + t < n ? (uint64_t)tab.c[tail[(size_t)TAILB - 1 - j + t]] : 0ull;
                w1 = (ks.lc >= 64 ? 0ull : (w1 << ks.lc)) | code;
            }
            const uint64_t prefix = lo1 >= 64 ? 0ull : (w1 >> lo1);
            short_words[(size_t)((prefix >> low) & (RADIX - 1))].push_back((prefix << sfield) | pos);
        }
    }
    // the buckets' sizes and their owners (multi_plan.hpp: deal_top_digit_buckets -- dealt whole, in order; rank d starts at the first
    // bucket boundary at or behind the start of its block, so it holds the tail of its block plus a little and receives the head in
    // front of its own records: re-balance in place)
    int QR = solo_ ? 1 : 4;
    if (pieces_env_ > 0) QR = std::max(1, std::min(16, pieces_env_));
    std::vector<uint64_t> shorts(RADIX, 0);
    for (int b = 0; b < RADIX; ++b) shorts[b] = short_words[b].size();
    const plan::OneWordDeal deal = plan::deal_top_digit_buckets(table.data(), W, P, shorts, targets, trust, !solo_, QR);
    if (deal.PT[RADIX] != n) { mg_set_err(g, "one-word first sort: the top-digit counts do not add up to the text"); return PSACX_EDEVICE; }
    if (!deal.ok) return PSACX_RETRY_;
    // (the scratch of the bucket passes was sized above for the tiles of a share of 9/8 of a block and 64 more per bucket: a deal forced beyond
    //  that -- `trust` -- is turned away here, before anything has moved, instead of failing on the receiving side)
    for (int r = 0; r < P; ++r) if (deal.cs[r] > sizes[r] + sizes[r] / 8 + 256 + (uint64_t)RADIX * 63 * TILE) return PSACX_RETRY_;
    const std::vector<uint64_t>& tot = deal.tot; const std::vector<uint64_t>& PT = deal.PT;
    const std::vector<int>& cut = deal.cut;
    const std::vector<uint64_t>& Gs = deal.Gs; const std::vector<uint64_t>& cs = deal.cs; const std::vector<uint64_t>& Hs = deal.Hs;
    const std::vector<uint64_t>& rooms = deal.rooms;
    const bool inplace = deal.inplace;
    (void)PT;
    *lo1_out = lo1;
    // ---- 3. arrays: the partitioned block (grp), two record arrays of the rank's share (A, B) and the suffixes of the last pass (vout).
    //      Reduced-memory layout: grp, the array that does not end up with word 1 and the suffixes are the rank's three output arrays.
    const int npass = (int)((low + RADIX_BITS - 1) / RADIX_BITS);
    std::vector<DBuf<T>> grp(L), A(L), B(L), vout(L);
    std::vector<DBuf<uint8_t>> dig(L);          // the digit the next bucket pass sorts on, a byte per record (engine.hpp: onew_bucket_passes)
    std::vector<uint64_t> share(L);
    int rc_alloc = PSACX_OK;
    for (int i = 0; i < L && rc_alloc == PSACX_OK; ++i) {
        psacx_ctx* c = ctx(i);
        const int me = rank(i);
        drop3(i, rec[i]);
        share[i] = cs[me];
        const uint64_t ng = solo_ ? S[i].m : nrec[i], room = rooms[me];
        const bool lend = diet && !S[i].out_busy && std::max(room, ng) <= S[i].out_cap;
        DBuf<T>& k1_final = (npass & 1) ? B[i] : A[i];          // the array the last pass writes word 1 into
        DBuf<T>& other = (npass & 1) ? A[i] : B[i];
        if (lend) {
            S[i].out_busy = true;
            other.borrow(c, S[i].ISA, room);
            vout[i].borrow(c, S[i].SA, room);
            if (S[i].LCP && !solo_) { grp[i].borrow(c, S[i].LCP, ng); invalidate_lcp_pyramid(i); }
        } else {
            rc_alloc = other.alloc(c, room, reserve_of(i));
            if (rc_alloc == PSACX_OK) rc_alloc = vout[i].alloc(c, room, reserve_of(i));
        }
        if (rc_alloc == PSACX_OK) rc_alloc = k1_final.alloc(c, room, reserve_of(i));
        if (rc_alloc == PSACX_OK && !solo_ && !grp[i].p) rc_alloc = grp[i].alloc(c, ng, reserve_of(i));
        if (rc_alloc == PSACX_OK && npass > 1) rc_alloc = dig[i].alloc(c, room + 64);
        if (rc_alloc != PSACX_OK) mg_set_err(g, "one-word first sort: record arrays: " + c->hip_err);
    }
    PSACX_TRY(agree(rc_alloc));
    // ---- 4. the pass on the top digit, word 1 computed on the spot (one rank without the wire: straight into A)
    PSACX_TRY(par([&](int i) -> int {
        psacx_ctx* c = ctx(i);
        Scr& q = scr[i];
        if (!q.ntiles) return PSACX_OK;
        MG_HIP(g, hipSetDevice(c->device));
        MG_HIP(g, hipMemsetAsync(q.desc, 0, 256, c->stream));
        const uint64_t cnt = solo_ ? S[i].m : nrec[i];
        hipLaunchKernelGGL((key_scatter1w_kernel<BLOCK, ITEMS>), dim3((unsigned)q.ntiles), dim3(BLOCK), 0, c->stream, (const uint8_t*)tbuf[i].p, cnt, S[i].m + two_k, tab, ks0,
                           reinterpret_cast<uint64_t*>(solo_ ? A[i].p : grp[i].p), (int)(lo1 + low), q.base0, q.tile_hist0, q.slab_tot0, reinterpret_cast<unsigned*>(q.desc),
                           sort_chunk_for(cnt, true), q.slab0, lo1 | (sfield << 16), solo_ ? (uint64_t)0 : S[i].off);
        MG_HIP(g, hipGetLastError());
        return PSACX_OK;
    }));
    mark("    sort: keys + partition by the top digit");
    // ---- 5. where everything lands: bucket b of rank `me` = [short suffixes][sender 0] .. [sender P - 1]
    std::vector<std::vector<uint64_t>> boff(L, std::vector<uint64_t>(RADIX + 1, 0));       // start of bucket b in the rank's arrays
    for (int i = 0; i < L; ++i) {
        const int me = rank(i);
        uint64_t at = Hs[me];
        for (int b = 0; b <= RADIX; ++b) { boff[i][b] = at; if (b < RADIX && b >= cut[me] && b < cut[me + 1]) at += tot[b]; }
    }
    const std::vector<std::vector<int>>& rcuts = deal.rcuts;       // the buckets of a destination in QR ranges of about equal size
    std::vector<std::vector<hipEvent_t>> done(QR, std::vector<hipEvent_t>(L, nullptr));
    auto drop_events = [&]() { for (auto& v : done) for (int i = 0; i < L; ++i) if (v[i]) { (void)hipSetDevice(ctx(i)->device); (void)hipEventDestroy(v[i]); v[i] = nullptr; } };
    int rc = PSACX_OK;
    if (!solo_) {
        for (int q = 0; q < QR && rc == PSACX_OK; ++q) for (int i = 0; i < L && rc == PSACX_OK; ++i) {
            if (hipSetDevice(ctx(i)->device) != hipSuccess || hipEventCreateWithFlags(&done[q][i], hipEventDisableTiming) != hipSuccess) { mg_set_err(g, "one-word first sort: event creation failed"); rc = PSACX_EHIP; }
        }
        // the short suffixes at the head of their buckets (before the first exchange is issued: the copies are ordered on the compute streams,
        // which the range sorts wait on anyway)
        for (int i = 0; i < L && rc == PSACX_OK; ++i) {
            const int me = rank(i);
            (void)hipSetDevice(ctx(i)->device);
            for (int b = cut[me]; b < cut[me + 1] && rc == PSACX_OK; ++b)
                if (!short_words[b].empty() && hipMemcpyAsync(A[i].p + boff[i][b], short_words[b].data(), short_words[b].size() * 8, hipMemcpyHostToDevice, ctx(i)->stream) != hipSuccess) {
                    mg_set_err(g, "one-word first sort: copy of the short suffixes failed"); rc = PSACX_EHIP;
                }
        }
        // the messages from sender r to destination d in range q (multi_plan.hpp: OneWordDeal::pieces; sender and receiver derive their
        // lists from that one function)
        typedef plan::Piece Piece;
        auto pieces = [&](int r, int d, int q) -> std::vector<Piece> { return deal.pieces(r, d, q); };
        for (int q = 0; q < QR && rc == PSACX_OK; ++q) {
            std::vector<std::vector<Msg>> sends(L), recvs(L);
            std::vector<std::vector<const void*>> in(L);
            std::vector<std::vector<void*>> out(L);
            for (int i = 0; i < L; ++i) {
                const int me = rank(i);
                for (int d = 0; d < P; ++d) for (const Piece& pc : pieces(me, d, q)) sends[i].push_back(Msg{d, pc.soff, pc.cnt});
                for (int r = 0; r < P; ++r) for (const Piece& pc : pieces(r, me, q)) recvs[i].push_back(Msg{r, pc.roff, pc.cnt});
                in[i] = {grp[i].p}; out[i] = {A[i].p};
            }
            rc = transfer(in, out, {sizeof(T)}, sends, recvs, &done[q]);
        }
    }
    // ---- 6. the LSD passes inside the buckets of a range as soon as it has landed
    std::vector<std::vector<std::vector<unsigned long long>>> tabs(L, std::vector<std::vector<unsigned long long>>(QR));
    std::vector<uint64_t*> s1(L, nullptr);
    for (int q = 0; q < QR && rc == PSACX_OK; ++q) {
        rc = par([&](int i) -> int {
            psacx_ctx* c = ctx(i);
            MG_HIP(g, hipSetDevice(c->device));
            if (!solo_) for (int s2 = 0; s2 < L; ++s2) MG_HIP(g, hipStreamWaitEvent(c->stream, done[q][s2], 0));
            std::vector<unsigned long long>& ht = tabs[i][q];
            ht.assign(2 * (RADIX + 1), 0);
            const int b0 = rcuts[rank(i)][q], b1 = rcuts[rank(i)][q + 1];
            uint64_t cntq = 0;
            for (int b = 0; b <= RADIX; ++b) ht[b] = boff[i][std::min(std::max(b, b0), b1)];
            cntq = ht[RADIX] - ht[0];
            if (!cntq) { if (!s1[i]) s1[i] = reinterpret_cast<uint64_t*>(((npass & 1) ? B[i] : A[i]).p); return PSACX_OK; }
            const OneWordLayout lay = onew_layout<TILE>(ht.data(), (share[i] + TILE - 1) / TILE);
            if (lay.need > scr[i].desc_bytes || lay.vtiles >= (1ull << 31)) { mg_set_err(g, "one-word first sort: scratch of the bucket passes too small"); return PSACX_EDEVICE; }
            uint64_t* res = nullptr;
            MG_OP(g, c, onew_bucket_passes(c, scr[i].desc, ht.data(), lay, reinterpret_cast<uint64_t*>(A[i].p), reinterpret_cast<uint64_t*>(B[i].p),
                                           reinterpret_cast<uint64_t*>(vout[i].p), sfield, low, lo1, cntq, &res, nullptr, dig[i].p));
            s1[i] = res;
            return PSACX_OK;
        });
    }
    // everything has arrived and every pass has run before the partitioned blocks and the tables go away
    for (int i = 0; i < L; ++i) {
        (void)hipSetDevice(ctx(i)->device);
        if (!solo_) for (int q = 0; q < QR; ++q) for (int s2 = 0; s2 < L; ++s2) if (done[q][s2]) (void)hipStreamWaitEvent(ctx(i)->stream, done[q][s2], 0);
    }
    for (int i = 0; i < L; ++i) { (void)hipSetDevice(ctx(i)->device); (void)hipStreamSynchronize(ctx(i)->stream); }
    drop_events();
    PSACX_TRY(agree(rc));
    for (int i = 0; i < L; ++i) {
        DBuf<T>& k1_final = (npass & 1) ? B[i] : A[i];
        if (s1[i] && reinterpret_cast<T*>(s1[i]) != k1_final.p) { mg_set_err(g, "one-word first sort: word 1 ended in the wrong array"); return PSACX_EDEVICE; }
        grp[i].release(); dig[i].release();
        ((npass & 1) ? A[i] : B[i]).release();
        rec[i] = Rec<T>();
        rec[i].k1 = std::move(k1_final); rec[i].v = std::move(vout[i]); rec[i].cnt = share[i];
        rec[i].k1.advance(Hs[rank(i)]); rec[i].v.advance(Hs[rank(i)]);
        rec[i].k1.n = share[i]; rec[i].v.n = share[i];
    }
    if (inplace) {
        head_.assign(L, 0); room_.assign(L, 0);
        for (int i = 0; i < L; ++i) { head_[i] = Hs[rank(i)]; room_[i] = rooms[rank(i)]; }
        held_from_ = Gs; held_cnt_ = cs;
    }
    g->last_one_word = true;
    mark("    sort: shuffle by buckets + bucket passes");
    const int rct = first_sort_ties(rec, targets, bits1, bits2, lo1, tbuf, two_k, tab, ks, true);
    head_.clear(); room_.clear();
    return rct;
    }
}

} // namespace psacx
