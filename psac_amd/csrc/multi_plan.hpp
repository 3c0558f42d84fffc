// multi_plan.hpp -- the planning of the block-distributed construction that needs no device: who owns what, who sends what to
// whom and where it lands.  Plain C++ (no HIP call, no device pointer), so that it is compiled and exercised without a GPU
// (tests/cpp/test_multi_plan.cpp plays the exchanges it plans on host arrays); multi.hpp executes the plans.
//
//   BlkDist                  mxx::blk_dist as psac uses it (suffix_array.hpp:183-194, 221; bulk_permute.hpp:23)
//   sample_positions,        the splitters of the sample sort that stands in for mxx::sort (idxsort.hpp:60-62): one sample per stratum,
//   choose_splitters,        ties divided by (rank, index); the destination of a record is the number of splitters that do not sort
//   destination_of           after it
//   rebalance_bounds         exact re-balance of globally sorted records to the block sizes (the per-rank counts mxx::sort preserves)
//   OneWordDeal              first round in one-word records: the 256 buckets of the top digit dealt whole to the ranks from exact
//                            counts, the messages of every (sender, destination, range) and the in-place re-balance afterwards
//   SliceShape               SA -> ISA by destination slices (bulk_permute.hpp:14-73): slice / window / level widths
#pragma once
#include <algorithm>
#include <cstdint>
#include <vector>

#if defined(__HIPCC__)
#define PSACX_HD __host__ __device__
#else
#define PSACX_HD
#endif

namespace psacx {

struct BlkDist {       // mxx::blk_dist: the first n mod P ranks hold one element more
    uint64_t n, div, mod; unsigned P;
    PSACX_HD unsigned rank_of(uint64_t g) const {
        const uint64_t big = (div + 1) * mod;
        if (g < big) return (unsigned)(g / (div + 1));
        return (unsigned)(mod + (g - big) / (div ? div : 1));
    }
    PSACX_HD uint64_t off(unsigned r) const { return div * r + (r < mod ? r : mod); }
    PSACX_HD uint64_t size(unsigned r) const { return div + (r < mod ? 1 : 0); }
};
inline BlkDist make_dist(uint64_t n, unsigned P) { BlkDist d; d.n = n; d.P = P; d.div = n / P; d.mod = n % P; return d; }

namespace plan {

struct Msg { int peer; uint64_t off, cnt; };      // `cnt` elements at element offset `off` of the local array, to / from rank `peer`

inline std::vector<uint64_t> prefix_of(const std::vector<uint64_t>& x) {
    std::vector<uint64_t> o(x.size() + 1, 0);
    for (size_t i = 0; i < x.size(); ++i) o[i + 1] = o[i] + x[i];
    return o;
}

// suffix_array.hpp:226-227: the blocks a caller hands over must be those of mxx::blk_dist
inline bool follows_blk_dist(const std::vector<uint64_t>& sizes) {
    uint64_t n = 0;
    for (uint64_t s : sizes) n += s;
    const uint64_t P = sizes.size();
    for (uint64_t r = 0; r < P; ++r) if (sizes[r] != n / P + (r < n % P ? 1 : 0)) return false;
    return true;
}

// ---------------------------------------------------------------- sample sort
// One sample from a pseudo-random place in each of `samples` equal strata of a rank's `cnt` records.  (Evenly spaced samples alias
// with periodic text: in a tandem repeat whose period divides the spacing every sample of every rank carries the same key.)
inline std::vector<uint64_t> sample_positions(uint64_t cnt, int rank, uint64_t call, int samples) {
    std::vector<uint64_t> pos;
    for (int s = 0; s < samples && cnt; ++s) {
        const uint64_t lo = (uint64_t)(((unsigned __int128)cnt * s) / samples), hi = (uint64_t)(((unsigned __int128)cnt * (s + 1)) / samples);
        if (hi <= lo) continue;
        uint64_t z = ((uint64_t)rank << 32 | (uint64_t)s) + 0x9E3779B97F4A7C15ull * (call + 1);      // splitmix64
        z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull; z = (z ^ (z >> 27)) * 0x94D049BB133111EBull; z ^= z >> 31;
        const uint64_t p = lo + z % (hi - lo);
        if (pos.empty() || pos.back() != p) pos.push_back(p);
    }
    return pos;
}

struct Smp {       // a sampled record in the total order (k1, k2, rank, index): equal keys are divided between the ranks
    uint64_t k1, k2, r, p;
    bool operator<(const Smp& o) const { return k1 != o.k1 ? k1 < o.k1 : k2 != o.k2 ? k2 < o.k2 : r != o.r ? r < o.r : p < o.p; }
    bool operator==(const Smp& o) const { return k1 == o.k1 && k2 == o.k2 && r == o.r && p == o.p; }
};

// P - 1 splitters at the P-quantiles of all samples (sorted, duplicates dropped: at most P - 1 come back)
inline std::vector<Smp> choose_splitters(std::vector<Smp> flat, int P) {
    std::sort(flat.begin(), flat.end());
    std::vector<Smp> spl;
    for (int d = 1; d < P && !flat.empty(); ++d) spl.push_back(flat[std::min(flat.size() - 1, flat.size() * (size_t)d / (size_t)P)]);
    std::sort(spl.begin(), spl.end());
    spl.erase(std::unique(spl.begin(), spl.end()), spl.end());
    return spl;
}

// destination of record (k1, k2) at index idx of rank `rank`: the number of splitters that do not sort after it (radix.hpp:
// classify_kernel computes the same on the device)
inline unsigned destination_of(const std::vector<Smp>& spl, uint64_t k1, uint64_t k2, uint64_t rank, uint64_t idx) {
    const Smp me{k1, k2, rank, idx};
    unsigned lo = 0, hi = (unsigned)spl.size();
    while (lo < hi) { const unsigned mid = (lo + hi) >> 1; if (me < spl[mid]) hi = mid; else lo = mid + 1; }
    return lo;
}

// A rank holds the globally sorted records g_start .. g_start + cnt and the blocks start at TP[0 .. P]: bounds[d] = local index of the
// first record that belongs to rank d (bounds[P] = cnt)
inline std::vector<uint64_t> rebalance_bounds(uint64_t g_start, uint64_t cnt, const std::vector<uint64_t>& TP) {
    const size_t P = TP.size() - 1;
    std::vector<uint64_t> b(P + 1, cnt);
    for (size_t d = 0; d < P; ++d) b[d] = std::min<uint64_t>(TP[d] > g_start ? TP[d] - g_start : 0, cnt);
    return b;
}

// ---------------------------------------------------------------- first round in one-word records (multi.hpp: sort_first_one_word)
// table[r * W + b]: records of top digit b on sender r (W >= 256); shorts[b]: the suffixes shorter than the window that belong to bucket b
// (made on the host, placed at the head of the bucket); targets: the block sizes.
constexpr int DEAL_BUCKETS = 256;
struct Piece { uint64_t soff, roff, cnt; };       // sender offset in its partitioned block, receiver offset in its record array
struct OneWordDeal {
    int P = 0, QR = 1;
    bool ok = false;               // false: the buckets cannot be dealt within the slack of the record arrays (and the caller did not insist)
    bool inplace = false;          // every rank holds the tail of its own block: re-balance in place
    std::vector<uint64_t> tot, PT;         // size of every bucket, its global start (PT[256] = n)
    std::vector<int> cut;                  // rank d owns the buckets cut[d] .. cut[d + 1] - 1
    std::vector<uint64_t> Gs, cs, Hs, rooms;   // per rank: global index of its first record, its share, headroom in front of it, room of its arrays
    std::vector<std::vector<int>> rcuts;   // per destination: QR + 1 bucket cuts of its ranges
    std::vector<uint64_t> shorts;
    const uint64_t* table = nullptr; int W = 0;

    // start of bucket b in rank d's arrays (b = 256: the end of its share)
    uint64_t bucket_start(int d, int b) const {
        uint64_t at = Hs[d];
        for (int x = cut[d]; x < b && x < cut[d + 1]; ++x) at += tot[x];
        return at;
    }
    // the messages from sender r to destination d in range q: one per bucket, neighbours joined where they are contiguous on both sides
    // (always on the sender's; on the receiver's when no other sender's records and no short suffix lie between them).  Sender and
    // receiver derive their lists from this one function.
    std::vector<Piece> pieces(int r, int d, int q) const {
        std::vector<Piece> out;
        uint64_t so = 0;
        for (int b = 0; b < rcuts[d][q]; ++b) so += table[(size_t)r * W + b];
        uint64_t bs = bucket_start(d, rcuts[d][q]);
        for (int b = rcuts[d][q]; b < rcuts[d][q + 1]; ++b) {
            const uint64_t cn = table[(size_t)r * W + b];
            uint64_t ro = bs + shorts[b];
            for (int r2 = 0; r2 < r; ++r2) ro += table[(size_t)r2 * W + b];
            if (cn) {
                if (!out.empty() && out.back().soff + out.back().cnt == so && out.back().roff + out.back().cnt == ro) out.back().cnt += cn;
                else out.push_back(Piece{so, ro, cn});
            }
            so += cn;
            bs += tot[b];
        }
        return out;
    }
};

// Buckets are dealt whole, in order.  Rank d starts at the first bucket boundary at or behind the start of its block: every rank then
// holds a little more than the tail of its own block -- the head, at most one bucket, sits at the end of the rank before it and is
// received in front of the rank's own records, for which the arrays leave room (Hs / rooms; in_place_messages below): no copy of the
// record arrays to re-balance them.  slack_div: the arrays hold block + block / slack_div records.  trust: deal anyhow (tests); the
// re-balance then goes through a copy (inplace = false).  with_wire: false for one rank that keeps everything (no headroom needed).
inline OneWordDeal deal_top_digit_buckets(const uint64_t* table, int W, int P, const std::vector<uint64_t>& shorts, const std::vector<uint64_t>& targets,
                                          bool trust, bool with_wire, int QR, uint64_t slack_div = 8) {
    OneWordDeal D;
    D.P = P; D.QR = QR; D.table = table; D.W = W; D.shorts = shorts;
    D.tot.assign(DEAL_BUCKETS, 0); D.PT.assign(DEAL_BUCKETS + 1, 0);
    for (int b = 0; b < DEAL_BUCKETS; ++b) {
        D.tot[b] = shorts[b];
        for (int r = 0; r < P; ++r) D.tot[b] += table[(size_t)r * W + b];
        D.PT[b + 1] = D.PT[b] + D.tot[b];
    }
    const std::vector<uint64_t> TP = prefix_of(targets);
    D.cut.assign(P + 1, 0);
    D.cut[P] = DEAL_BUCKETS;
    for (int d = 1; d < P; ++d) {
        int b = D.cut[d - 1];
        while (b < DEAL_BUCKETS && D.PT[b] < TP[d]) ++b;
        D.cut[d] = b;
    }
    D.Gs.assign(P, 0); D.cs.assign(P, 0); D.Hs.assign(P, 0); D.rooms.assign(P, 0);
    D.inplace = with_wire;
    D.ok = true;
    for (int d = 0; d < P; ++d) {
        D.Gs[d] = D.PT[D.cut[d]]; D.cs[d] = D.PT[D.cut[d + 1]] - D.PT[D.cut[d]];
        D.Hs[d] = D.Gs[d] - TP[d]; D.rooms[d] = std::max(D.Hs[d] + D.cs[d], targets[d]);
        if (D.rooms[d] > targets[d] + targets[d] / slack_div) {       // (the slack of the reduced-memory layout's record arrays)
            if (!trust) { D.ok = false; return D; }
            D.inplace = false;
        }
    }
    if (!D.inplace) for (int d = 0; d < P; ++d) { D.Hs[d] = 0; D.rooms[d] = D.cs[d]; }
    // the buckets of a destination in QR ranges of about equal size (the same cuts on every rank: a sender must know the ranges of its
    // destinations)
    D.rcuts.assign(P, std::vector<int>());
    for (int d = 0; d < P; ++d) {
        const int nb = D.cut[d + 1] - D.cut[d];
        const int qr = std::max(1, std::min(QR, nb));
        const uint64_t sh = D.PT[D.cut[d + 1]] - D.PT[D.cut[d]];
        std::vector<int> rc(QR + 1, D.cut[d + 1]);
        rc[0] = D.cut[d];
        for (int q = 1; q < qr; ++q) {
            int b = rc[q - 1];
            const uint64_t want = D.PT[D.cut[d]] + (uint64_t)(((unsigned __int128)sh * q) / qr);
            while (b < D.cut[d + 1] && D.PT[b + 1] <= want) ++b;
            rc[q] = std::max(b, rc[q - 1]);
        }
        D.rcuts[d] = rc;
    }
    return D;
}

// Re-balance without a copy: rank `me` holds the globally sorted records held_from[me] .. + held_cnt[me] `head` elements into its arrays
// and its block [TP[me], TP[me + 1]) starts at most `head` records before them.  sends: what other ranks' blocks it holds (offsets in its
// arrays); recvs: the pieces of its block that others hold, to where the block will begin at the start of the arrays.  false: the rank
// does not hold the tail of its block (a plan that cannot be executed in place).
inline bool in_place_messages(int me, int P, const std::vector<uint64_t>& held_from, const std::vector<uint64_t>& held_cnt, const std::vector<uint64_t>& TP,
                              uint64_t head, std::vector<Msg>& sends, std::vector<Msg>& recvs) {
    const uint64_t g0 = held_from[me], g1 = g0 + held_cnt[me];
    if (g0 < TP[me] || g0 - TP[me] != head || g1 < TP[me + 1]) return false;
    for (int d = 0; d < P; ++d) {
        if (d == me) continue;
        const uint64_t lo = std::max(g0, TP[d]), hi = std::min(g1, TP[d + 1]);
        if (lo < hi) sends.push_back(Msg{d, head + (lo - g0), hi - lo});
    }
    for (int r = 0; r < P; ++r) {
        if (r == me) continue;
        const uint64_t lo = std::max(held_from[r], TP[me]), hi = std::min(held_from[r] + held_cnt[r], TP[me + 1]);
        if (lo < hi) recvs.push_back(Msg{r, lo - TP[me], hi - lo});
    }
    return true;
}

// ---------------------------------------------------------------- SA -> ISA by destination slices (multi.hpp: isa_by_slices_t)
// A block of max_m positions is cut into spo slices of 2^sb positions; a slice is partitioned further by `cbs` (bits per level, at most
// nine each) down to windows of 2^wb positions that are scattered inside LDS.  max_classes bounds P * spo (the histogram of the first
// level lives in LDS), wbmax the window (14 bits for 32-bit ranks, 13 for 64-bit ones).  env_wb / env_s1: test overrides (0 = none).
struct SliceShape { unsigned kb, s1, sb, spo, wb, rbits, levels2, C; std::vector<unsigned> cbs; };
inline SliceShape slice_shape(uint64_t max_m, unsigned P, unsigned wbmax_type, unsigned max_classes, unsigned env_wb, unsigned env_s1) {
    constexpr unsigned TILE_BITS = 13;
    auto bits_for_ = [](uint64_t v) { unsigned b = 0; while (b < 64 && (v >> b) != 0) ++b; return b ? b : 1u; };
    SliceShape S;
    S.kb = bits_for_(max_m > 1 ? max_m - 1 : 1);
    unsigned cap_bits = 0;
    while ((2u << cap_bits) * P <= max_classes) ++cap_bits;        // most slice bits with P * 2^bits classes
    unsigned wbmax = wbmax_type;
    if (env_wb) wbmax = std::min<unsigned>(wbmax_type, std::max(4u, env_wb));     // (tests: levels on small inputs)
    if (env_s1) cap_bits = std::min<unsigned>(cap_bits, env_s1);
    unsigned s1 = std::min<unsigned>(cap_bits, S.kb > wbmax ? S.kb - wbmax : 0);
    // a further level walks tiles of 2^13 pairs that must not straddle slices
    if (S.kb - s1 > wbmax && S.kb - s1 < TILE_BITS) s1 = S.kb > TILE_BITS ? S.kb - TILE_BITS : 0;
    S.s1 = s1;
    S.sb = S.kb - s1;
    S.spo = (unsigned)((max_m + (1ull << S.sb) - 1) >> S.sb);
    S.wb = std::min(S.sb, wbmax);
    S.rbits = S.sb - S.wb;
    // levels of at most 9 bits each; a level's parent buckets (2^(shift + cb) pairs) must hold whole tiles, which only binds the last
    // level when a test shrinks the windows below a tile
    S.cbs.clear();
    if (S.rbits) {
        const unsigned last_min = S.wb >= TILE_BITS ? 1u : std::min(S.rbits, TILE_BITS - S.wb);
        unsigned nl = (S.rbits + 8) / 9;
        S.cbs.assign(nl, 0);
        for (unsigned j = 0; j < nl; ++j) S.cbs[j] = S.rbits / nl + (j < S.rbits % nl ? 1 : 0);
        if (S.cbs.back() < last_min) {
            const unsigned rest = S.rbits - last_min;
            nl = 1 + (rest + 8) / 9;
            S.cbs.assign(nl, 0);
            for (unsigned j = 0; j + 1 < nl; ++j) S.cbs[j] = rest / (nl - 1) + (j < rest % (nl - 1) ? 1 : 0);
            S.cbs.back() = last_min;
        }
    }
    S.levels2 = (unsigned)S.cbs.size();
    S.C = P * S.spo;
    return S;
}

// slices per step of the owner-side levels: all of a block at once in the normal layout, about an eighth of the block in the reduced one
inline uint64_t slices_per_step(const SliceShape& S, uint64_t max_m, bool reduced, uint64_t env_step) {
    uint64_t G = S.spo;
    if (reduced) G = std::max<uint64_t>(1, std::max<uint64_t>(1ull << S.sb, max_m / 8) >> S.sb);
    if (env_step) G = env_step;
    return std::min<uint64_t>(G, S.spo);
}

} // namespace plan
} // namespace psacx
