// C interface over psac_amd/csrc/multi_plan.hpp for tests/test_dist_cpu.py: two (or more) real processes exchange numpy arrays over gloo
// exactly as the shipped planning code says -- the same functions multi.hpp calls before it issues ncclSend / ncclRecv.  Built by g++.
#include <cstring>
#include "../../psac_amd/csrc/multi_plan.hpp"

using namespace psacx;
using namespace psacx::plan;

static OneWordDeal make_deal(const uint64_t* table, int P, const uint64_t* shorts, const uint64_t* targets, int trust, int QR) {
    std::vector<uint64_t> sh(shorts, shorts + 256), tg(targets, targets + P);
    return deal_top_digit_buckets(table, 256, P, sh, tg, trust != 0, P > 1, QR);
}

extern "C" {

// 0: dealt; 1: refused (a bucket cannot be dealt within the slack).  cut[P + 1]; Gs, cs, Hs, rooms [P]
int plan_deal(const uint64_t* table, int P, const uint64_t* shorts, const uint64_t* targets, int trust, int QR, int* cut, uint64_t* Gs, uint64_t* cs,
              uint64_t* Hs, uint64_t* rooms, int* inplace) {
    const OneWordDeal D = make_deal(table, P, shorts, targets, trust, QR);
    if (!D.ok) return 1;
    for (int d = 0; d <= P; ++d) cut[d] = D.cut[d];
    for (int d = 0; d < P; ++d) { Gs[d] = D.Gs[d]; cs[d] = D.cs[d]; Hs[d] = D.Hs[d]; rooms[d] = D.rooms[d]; }
    *inplace = D.inplace ? 1 : 0;
    return 0;
}
// start of bucket b in rank d's arrays
uint64_t plan_bucket_start(const uint64_t* table, int P, const uint64_t* shorts, const uint64_t* targets, int trust, int QR, int d, int b) {
    return make_deal(table, P, shorts, targets, trust, QR).bucket_start(d, b);
}
// the messages from sender r to destination d in range q; returns their number (at most 256)
int plan_pieces(const uint64_t* table, int P, const uint64_t* shorts, const uint64_t* targets, int trust, int QR, int r, int d, int q, uint64_t* soff, uint64_t* roff,
                uint64_t* cnt) {
    const OneWordDeal D = make_deal(table, P, shorts, targets, trust, QR);
    const std::vector<Piece> pc = D.pieces(r, d, q);
    for (size_t i = 0; i < pc.size(); ++i) { soff[i] = pc[i].soff; roff[i] = pc[i].roff; cnt[i] = pc[i].cnt; }
    return (int)pc.size();
}
// in-place re-balance of rank `me`: sends / recvs as (peer, offset, count) triples; returns 0, or 1 when the rank does not hold the tail of its block
int plan_in_place(int me, int P, const uint64_t* held_from, const uint64_t* held_cnt, const uint64_t* TP, uint64_t head, int64_t* sends, int* ns, int64_t* recvs, int* nr) {
    std::vector<uint64_t> hf(held_from, held_from + P), hc(held_cnt, held_cnt + P), tp(TP, TP + P + 1);
    std::vector<Msg> s, r;
    if (!in_place_messages(me, P, hf, hc, tp, head, s, r)) return 1;
    *ns = (int)s.size(); *nr = (int)r.size();
    for (size_t i = 0; i < s.size(); ++i) { sends[3 * i] = s[i].peer; sends[3 * i + 1] = (int64_t)s[i].off; sends[3 * i + 2] = (int64_t)s[i].cnt; }
    for (size_t i = 0; i < r.size(); ++i) { recvs[3 * i] = r[i].peer; recvs[3 * i + 1] = (int64_t)r[i].off; recvs[3 * i + 2] = (int64_t)r[i].cnt; }
    return 0;
}
// sample sort: positions of a rank's samples; returns their number
int plan_sample_positions(uint64_t cnt, int rank, uint64_t call, int samples, uint64_t* pos) {
    const std::vector<uint64_t> p = sample_positions(cnt, rank, call, samples);
    for (size_t i = 0; i < p.size(); ++i) pos[i] = p[i];
    return (int)p.size();
}
// splitters from all samples (k1, k2, rank, index per sample); returns their number (< P); out: 4 words per splitter
int plan_splitters(const uint64_t* samples, int nsamples, int P, uint64_t* out) {
    std::vector<Smp> flat;
    for (int i = 0; i < nsamples; ++i) flat.push_back(Smp{samples[4 * i], samples[4 * i + 1], samples[4 * i + 2], samples[4 * i + 3]});
    const std::vector<Smp> spl = choose_splitters(flat, P);
    for (size_t i = 0; i < spl.size(); ++i) { out[4 * i] = spl[i].k1; out[4 * i + 1] = spl[i].k2; out[4 * i + 2] = spl[i].r; out[4 * i + 3] = spl[i].p; }
    return (int)spl.size();
}
void plan_destinations(const uint64_t* splitters, int ns, const uint64_t* k1, const uint64_t* k2, uint64_t cnt, uint64_t rank, uint32_t* dest) {
    std::vector<Smp> spl;
    for (int i = 0; i < ns; ++i) spl.push_back(Smp{splitters[4 * i], splitters[4 * i + 1], splitters[4 * i + 2], splitters[4 * i + 3]});
    for (uint64_t i = 0; i < cnt; ++i) dest[i] = destination_of(spl, k1[i], k2[i], rank, i);
}
void plan_rebalance_bounds(uint64_t g_start, uint64_t cnt, const uint64_t* TP, int P, uint64_t* bounds) {
    std::vector<uint64_t> tp(TP, TP + P + 1);
    const std::vector<uint64_t> b = rebalance_bounds(g_start, cnt, tp);
    for (int d = 0; d <= P; ++d) bounds[d] = b[d];
}
void plan_blk(uint64_t n, unsigned P, uint64_t* sizes) { const BlkDist d = make_dist(n, P); for (unsigned r = 0; r < P; ++r) sizes[r] = d.size(r); }

}
