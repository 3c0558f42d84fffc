// The host-side plans of the multi-GPU engine (psac_amd/csrc/multi_plan.hpp), exercised WITHOUT a GPU: this file is compiled by g++
// (no hipcc) and plays every exchange a plan describes on plain host arrays.  The product (multi.hpp) executes the same plans with
// ncclSend / ncclRecv; here a "message" is a copy between two std::vectors.
//   blk       mxx::blk_dist: sizes / offsets / rank_of agree, printed for the Python side to compare with its own blk_sizes
//   deal      first round in one-word records: buckets of the top digit dealt whole; all pieces of all ranges land so that every
//             rank's share is ordered by (bucket, short suffixes first, sender, sequence); then the in-place re-balance leaves every
//             rank with exactly its block of the global order
//   sort      sample sort: splitters from the samples, destination of every record, local order, exact re-balance = the global order
//   slices    slice shapes: invariants for many block sizes and rank counts
// Exit code 0 and "ok" lines when everything holds; the first violation is printed and the exit code is 1.
#include <cstdio>
#include <cstdlib>
#include <cmath>
#include <cstring>
#include <numeric>
#include <random>
#include <string>
#include "../../psac_amd/csrc/multi_plan.hpp"

using namespace psacx;
using namespace psacx::plan;

static int fail(const std::string& m) { std::printf("FAIL %s\n", m.c_str()); return 1; }
#define CHECK(cond, msg) do { if (!(cond)) return fail(std::string(msg) + " [" #cond "]"); } while (0)

static int test_blk(uint64_t n, unsigned P) {
    const BlkDist d = make_dist(n, P);
    std::vector<uint64_t> sizes(P);
    uint64_t off = 0;
    for (unsigned r = 0; r < P; ++r) {
        sizes[r] = d.size(r);
        CHECK(d.off(r) == off, "blk_dist offset");
        for (uint64_t g = off; g < off + sizes[r]; g += std::max<uint64_t>(1, sizes[r] / 7)) CHECK(d.rank_of(g) == r, "blk_dist rank_of");
        if (sizes[r]) CHECK(d.rank_of(off + sizes[r] - 1) == r, "blk_dist rank_of (last)");
        off += sizes[r];
    }
    CHECK(off == n, "blk_dist total");
    CHECK(follows_blk_dist(sizes), "follows_blk_dist");
    if (P > 1 && n > P) { std::vector<uint64_t> bad = sizes; bad[0] += 1; bad[P - 1] -= 1; CHECK(!follows_blk_dist(bad), "follows_blk_dist rejects"); }
    std::printf("blk n=%llu P=%u sizes=", (unsigned long long)n, P);
    for (unsigned r = 0; r < P; ++r) std::printf("%s%llu", r ? "," : "", (unsigned long long)sizes[r]);
    std::printf("\n");
    return 0;
}

// ---- the one-word deal.  kind: 0 uniform digits, 1 geometric (skewed), 2 one bucket longer than a block, 3 few buckets (a tandem repeat)
struct RecId { int bucket; int sender; uint64_t seq; bool is_short; };
static bool rec_less(const RecId& a, const RecId& b) {
    if (a.bucket != b.bucket) return a.bucket < b.bucket;
    if (a.is_short != b.is_short) return a.is_short;          // the short suffixes stand at the head of their bucket
    if (a.sender != b.sender) return a.sender < b.sender;
    return a.seq < b.seq;
}
static bool rec_eq(const RecId& a, const RecId& b) { return a.bucket == b.bucket && a.sender == b.sender && a.seq == b.seq && a.is_short == b.is_short; }

static int test_deal(int P, uint64_t m, int kind, int QR, bool trust, unsigned seed, bool expect_ok) {
    std::mt19937_64 rng(seed);
    const int W = 256 + 5;
    const uint64_t n = m * P + (uint64_t)(P > 2 ? 2 : 0);     // (the first n mod P blocks are one longer)
    const BlkDist dist = make_dist(n, (unsigned)P);
    std::vector<uint64_t> targets(P);
    for (int r = 0; r < P; ++r) targets[r] = dist.size(r);
    // every sender's records by top digit; the last 40 positions of the text are short suffixes (made by the host, not by a sender)
    const uint64_t spec = std::min<uint64_t>(41, n);
    std::vector<uint64_t> table((size_t)P * W, 0), shorts(256, 0);
    std::vector<double> w(256, 1.0);
    if (kind == 1) for (int b = 0; b < 256; ++b) w[b] = 1.0 + (b % 5 == 0 ? 3.0 : 0.0) + (b / 64) * 0.5;        // 1 : 5.5 between buckets, the largest below 1.2 % of the text
    if (kind == 2) w[100] = 90.0 * (256.0 / P);                 // about 90 / (90 + P) ... of the text in one bucket: longer than a block
    if (kind == 3) { for (int b = 0; b < 256; ++b) w[b] = 0; for (int t = 0; t < 12; ++t) w[(int)(rng() % 256)] += 1 + (double)(rng() % 3); }
    std::discrete_distribution<int> pick(w.begin(), w.end());
    for (int r = 0; r < P; ++r) {
        uint64_t cnt = targets[r];
        if (r == P - 1) cnt -= std::min<uint64_t>(spec, cnt);   // (the short suffixes sit in the last block here)
        for (uint64_t j = 0; j < cnt; ++j) table[(size_t)r * W + pick(rng)]++;
    }
    { uint64_t left = spec - (targets[P - 1] < spec ? spec - targets[P - 1] : 0); for (uint64_t j = 0; j < left; ++j) shorts[pick(rng)]++;
      uint64_t have = 0; for (int r = 0; r < P; ++r) for (int b = 0; b < 256; ++b) have += table[(size_t)r * W + b];
      for (int b = 0; b < 256; ++b) have += shorts[b];
      while (have < n) { shorts[pick(rng)]++; ++have; } }
    const OneWordDeal D = deal_top_digit_buckets(table.data(), W, P, shorts, targets, trust, P > 1, QR);
    CHECK(D.PT[256] == n, "deal: bucket sizes add up to the text");
    if (!expect_ok) { CHECK(!D.ok, "deal: expected a refusal (a bucket that cannot be dealt within the slack)"); return 0; }
    if (!D.ok) std::printf("(P=%d kind=%d QR=%d) ", P, kind, QR);
    CHECK(D.ok, "deal: refused");
    // owners: every bucket has exactly one, in order; shares add up
    CHECK(D.cut[0] == 0 && D.cut[P] == 256, "deal: cuts cover all buckets");
    uint64_t sum = 0;
    for (int d = 0; d < P; ++d) { CHECK(D.cut[d] <= D.cut[d + 1], "deal: cuts ascend"); CHECK(D.Gs[d] == sum, "deal: a rank's first record"); sum += D.cs[d]; }
    CHECK(sum == n, "deal: shares add up");
    const std::vector<uint64_t> TP = prefix_of(targets);
    for (int d = 0; d < P; ++d) {
        if (D.inplace) { CHECK(D.Gs[d] >= TP[d] && D.Hs[d] == D.Gs[d] - TP[d], "deal: headroom = distance to the block start");
                         CHECK(D.rooms[d] <= targets[d] + targets[d] / 8, "deal: rooms within the slack"); }
        CHECK(D.rooms[d] >= D.Hs[d] + D.cs[d], "deal: room for the share behind its headroom");
    }
    // sender arrays: the partitioned block = records grouped by bucket, in sequence
    std::vector<std::vector<RecId>> sender(P);
    for (int r = 0; r < P; ++r) for (int b = 0; b < 256; ++b) for (uint64_t j = 0; j < table[(size_t)r * W + b]; ++j) sender[r].push_back(RecId{b, r, j, false});
    // receiver arrays with the short suffixes placed first (multi.hpp copies them to the head of their buckets)
    const RecId none{-1, -1, 0, false};
    std::vector<std::vector<RecId>> recv(P);
    for (int d = 0; d < P; ++d) {
        recv[d].assign(D.rooms[d], none);
        for (int b = D.cut[d]; b < D.cut[d + 1]; ++b) for (uint64_t j = 0; j < shorts[b]; ++j) recv[d][D.bucket_start(d, b) + j] = RecId{b, -1, j, true};
    }
    // all messages of all ranges; sender and receiver lists must agree (they come from one function) and nothing may be written twice
    for (int q = 0; q < QR; ++q) for (int r = 0; r < P; ++r) for (int d = 0; d < P; ++d)
        for (const Piece& pc : D.pieces(r, d, q)) {
            CHECK(pc.soff + pc.cnt <= sender[r].size(), "deal: a piece inside the sender's block");
            CHECK(pc.roff + pc.cnt <= recv[d].size(), "deal: a piece inside the receiver's arrays");
            for (uint64_t j = 0; j < pc.cnt; ++j) { CHECK(recv[d][pc.roff + j].bucket == -1, "deal: a slot is written once"); recv[d][pc.roff + j] = sender[r][pc.soff + j]; }
        }
    // every share: complete and in the order (bucket, shorts, sender, sequence) -- the order the stable LSD passes need
    std::vector<RecId> global;
    for (int d = 0; d < P; ++d) {
        for (uint64_t j = 0; j < D.Hs[d]; ++j) CHECK(recv[d][j].bucket == -1, "deal: the headroom stays empty");
        for (uint64_t j = 0; j < D.cs[d]; ++j) {
            const RecId& x = recv[d][D.Hs[d] + j];
            CHECK(x.bucket >= D.cut[d] && x.bucket < D.cut[d + 1], "deal: a record in its owner's share");
            if (j) CHECK(rec_less(recv[d][D.Hs[d] + j - 1], x), "deal: a share is in bucket / sender / sequence order");
            global.push_back(x);
        }
    }
    CHECK(global.size() == n, "deal: every record landed");
    for (size_t j = 1; j < global.size(); ++j) CHECK(rec_less(global[j - 1], global[j]), "deal: the shares concatenate to the global order");
    // the re-balance in place: every rank ends with its block of the global order at the start of its arrays
    if (D.inplace) {
        std::vector<std::vector<Msg>> sends(P), recvs(P);
        for (int me = 0; me < P; ++me) CHECK(in_place_messages(me, P, D.Gs, D.cs, TP, D.Hs[me], sends[me], recvs[me]), "in place: a rank holds the tail of its block");
        std::vector<std::vector<RecId>> after = recv;
        for (int me = 0; me < P; ++me) for (const Msg& rm : recvs[me]) {
            // the matching send of the peer: the same global records
            bool found = false;
            for (const Msg& sm : sends[rm.peer]) if (sm.peer == me && sm.cnt == rm.cnt) {
                const uint64_t gfirst = D.Gs[rm.peer] + (sm.off - D.Hs[rm.peer]);
                if (gfirst == TP[me] + rm.off) { for (uint64_t j = 0; j < rm.cnt; ++j) { CHECK(rm.off + j < after[me].size(), "in place: inside the arrays"); after[me][rm.off + j] = recv[rm.peer][sm.off + j]; } found = true; }
            }
            CHECK(found, "in place: every receive has its send");
        }
        for (int me = 0; me < P; ++me) for (uint64_t j = 0; j < targets[me]; ++j)
            CHECK(rec_eq(after[me][j], global[TP[me] + j]), "in place: a rank ends with its block of the global order");
    }
    return 0;
}

// ---- sample sort on host arrays
static int test_sort(int P, uint64_t m, int keyspace, unsigned seed) {
    std::mt19937_64 rng(seed);
    const uint64_t n = m * P + 1;
    const BlkDist dist = make_dist(n, (unsigned)P);
    std::vector<uint64_t> targets(P);
    for (int r = 0; r < P; ++r) targets[r] = dist.size(r);
    struct R { uint64_t k1, k2; int r; uint64_t idx; };
    std::vector<std::vector<R>> rec(P);
    for (int r = 0; r < P; ++r) for (uint64_t j = 0; j < targets[r]; ++j) rec[r].push_back(R{rng() % (uint64_t)keyspace, rng() % 3, r, j});
    const int SAMPLES = 64;
    std::vector<Smp> flat;
    for (int r = 0; r < P; ++r) {
        const std::vector<uint64_t> pos = sample_positions(rec[r].size(), r, 7, SAMPLES);
        for (size_t s = 1; s < pos.size(); ++s) CHECK(pos[s] > pos[s - 1], "samples: ascending, distinct");
        for (uint64_t p : pos) { CHECK(p < rec[r].size(), "samples: inside the block"); flat.push_back(Smp{rec[r][p].k1, rec[r][p].k2, (uint64_t)r, p}); }
    }
    const std::vector<Smp> spl = choose_splitters(flat, P);
    CHECK((int)spl.size() <= P - 1, "splitters: at most P - 1");
    for (size_t s = 1; s < spl.size(); ++s) CHECK(spl[s - 1] < spl[s], "splitters: strictly ascending");
    std::vector<std::vector<R>> got(P);
    for (int r = 0; r < P; ++r) for (const R& x : rec[r]) { const unsigned d = destination_of(spl, x.k1, x.k2, (uint64_t)x.r, x.idx); CHECK(d < (unsigned)P, "destination"); got[d].push_back(x); }
    auto less = [](const R& a, const R& b) { return a.k1 != b.k1 ? a.k1 < b.k1 : a.k2 < b.k2; };
    std::vector<uint64_t> counts(P);
    for (int d = 0; d < P; ++d) { std::stable_sort(got[d].begin(), got[d].end(), less); counts[d] = got[d].size(); }
    // destinations are monotone in the key order: the concatenation is sorted by (k1, k2)
    for (int d = 1; d < P; ++d) if (!got[d].empty()) for (int e = d - 1; e >= 0; --e) if (!got[e].empty()) { CHECK(!less(got[d].front(), got[e].back()), "sample sort: destinations ascend"); break; }
    // exact re-balance
    const std::vector<uint64_t> G = prefix_of(counts), TP = prefix_of(targets);
    std::vector<std::vector<R>> fin(P);
    for (int s = 0; s < P; ++s) {
        const std::vector<uint64_t> b = rebalance_bounds(G[s], counts[s], TP);
        CHECK(b[P] == counts[s], "rebalance bounds end at the count");
        for (int d = 0; d < P; ++d) { CHECK(b[d] <= b[d + 1], "rebalance bounds ascend"); for (uint64_t j = b[d]; j < b[d + 1]; ++j) fin[d].push_back(got[s][j]); }
    }
    for (int d = 0; d < P; ++d) { CHECK(fin[d].size() == targets[d], "rebalance: every rank ends with its block size"); for (size_t j = 1; j < fin[d].size(); ++j) CHECK(!less(fin[d][j], fin[d][j - 1]), "rebalance: sorted"); }
    for (int d = 1; d < P; ++d) if (!fin[d].empty() && !fin[d - 1].empty()) CHECK(!less(fin[d].front(), fin[d - 1].back()), "rebalance: blocks ascend");
    return 0;
}

static int test_slices() {
    for (unsigned P : {1u, 2u, 3u, 7u, 8u})
        for (unsigned wbmax : {13u, 14u})
            for (uint64_t m : {1ull, 100ull, 5003ull, 1ull << 14, (1ull << 20) + 17, 1ull << 28, (1ull << 32) - 1, 1ull << 32})
                for (unsigned env_wb : {0u, 4u, 6u}) for (unsigned env_s1 : {0u, 2u}) {
                    const SliceShape S = slice_shape(m, P, wbmax, 512, env_wb, env_s1);
                    CHECK(S.C == P * S.spo && S.C <= 512, "slices: classes fit the LDS histogram");
                    CHECK(((uint64_t)S.spo << S.sb) >= m && (S.spo == 1 || ((uint64_t)(S.spo - 1) << S.sb) < m), "slices: the slices cover the block exactly");
                    CHECK(S.wb <= wbmax && S.wb <= S.sb && S.rbits == S.sb - S.wb, "slices: window");
                    unsigned sum = 0;
                    for (unsigned c : S.cbs) { CHECK(c >= 1 && c <= 9, "slices: a level has 1 .. 9 bits"); sum += c; }
                    CHECK(sum == S.rbits && S.levels2 == S.cbs.size(), "slices: the levels take the slice down to windows");
                    // (the last level's parent buckets hold whole tiles of 2^13 pairs, or the whole slice is smaller than a tile)
                    if (S.levels2) CHECK(S.wb + S.cbs.back() >= 13 || S.sb < 13, "slices: the last level's buckets hold whole tiles");
                    for (bool red : {false, true}) { const uint64_t G = slices_per_step(S, m, red, 0); CHECK(G >= 1 && G <= S.spo, "slices: slices per step"); }
                }
    return 0;
}

int main() {
    int rc = 0;
    for (unsigned P : {1u, 2u, 3u, 7u, 8u}) for (uint64_t n : {1ull, 11ull, 1000ull, 4001ull, 1ull << 20}) rc |= test_blk(n, P);
    if (rc) return 1;
    std::printf("ok blk\n");
    for (int P : {2, 3, 7, 8}) {
        for (int QR : {1, 4, 7}) {
            rc |= test_deal(P, 40000, 0, QR, false, 1, true);        // uniform digits
            rc |= test_deal(P, 40000, 1, QR, false, 2, true);        // skewed digit counts
            rc |= test_deal(P, 40000, 3, QR, true, 3, true);         // a dozen buckets (tandem repeat), dealt anyhow
        }
        rc |= test_deal(P, 40000, 2, 4, false, 4, false);            // a bucket longer than a block: refused ...
        rc |= test_deal(P, 40000, 2, 4, true, 4, true);              // ... unless the caller insists: dealt, re-balanced through a copy
        if (rc) return 1;
    }
    rc |= test_deal(1, 5003, 0, 1, false, 5, true);
    std::printf("ok deal\n");
    for (int P : {2, 3, 7, 8}) for (int ks : {3, 1000, 1 << 30}) rc |= test_sort(P, 3000, ks, 11 + P);
    if (rc) return 1;
    std::printf("ok sort\n");
    rc |= test_slices();
    if (rc) return 1;
    std::printf("ok slices\n");
    return 0;
}
