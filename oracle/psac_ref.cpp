// oracle/psac_ref.cpp -- TEST INFRASTRUCTURE ONLY.
//
// CPU restatement (single rank, p = 1) of the SA / ISA / LCP construction path
// of patflick/psac, written from scratch for this repository.  It exists so
// that tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg can
// CHECK the HIP engine; nothing in psac_amd/ (the product) may link, import or
// call it.
//
// Parity pinning: psac itself (header-only C++ over the un-vendored, empty mxx
// submodule) cannot be built in this image.  Its own checker and CPU
// comparison, libdivsufsort, IS built from the reference tree into oracle/_ref/
// (oracle/Makefile: the four C files compiled where they lie, the two public
// headers instantiated from the reference's .h.cmake templates).  This
// restatement is pinned against (a) the reference's own known-answer vectors
// (test/test_psac.cpp:105 mississippi SA, test/test_bitops.cpp KATs,
// README.md print64 listing, test/test_gsa.cpp arrays, test/test_suffixtree.cpp
// table), (b) the SA/LCP/ISA checksums the survey captured from psac run in
// its container, committed under tests/golden/, and
// (c) outputs of the reference's libdivsufsort run here + Kasai (lcp.hpp:46-77)
// on the shapes of test/test_psac.cpp.  See tests/test_oracle_golden.py.
//
// The independent loops carry OpenMP pragmas; they only take effect in the second build
// (libpsac_oracle_mt.so: -fopenmp -D_GLIBCXX_PARALLEL, std::sort becomes the libstdc++ parallel
// sort), which bench.py's cpu_baseline leg times on all host cores.  The tests use the scalar build.
//
// Each function cites the reference file:line whose behaviour it follows.
// All citations are relative to /root/reference/.
#include <algorithm>
#include <cstdint>
#include <cstdlib>
#include <cstring>
#include <deque>
#include <limits>
#include <numeric>
#include <utility>
#include <vector>

namespace {

// ---------------------------------------------------------------- bit ops
// include/bitops.hpp:82-109 (leading_zeros: 64 for x == 0 at 64 bit, 32 at 32 bit)
inline unsigned clz64(uint64_t x) { return x ? (unsigned)__builtin_clzll(x) : 64u; }
inline unsigned clz32(uint32_t x) { return clz64((uint64_t)x) - 32u; }
template <typename T> inline unsigned lead_zeros(T x) {
    return sizeof(T) == 8 ? clz64((uint64_t)x) : clz32((uint32_t)x);
}
// include/bitops.hpp:35-58 (trailing zeros; undefined for 0 in the reference, 8*sizeof here)
template <typename T> inline unsigned trail_zeros(T x) {
    if (x == 0) return 8 * sizeof(T);
    return (unsigned)__builtin_ctzll((uint64_t)x);
}
// include/bitops.hpp:142-153
inline unsigned floor_log2(uint64_t n) { return 63u - (unsigned)__builtin_clzll(n); }
inline unsigned ceil_log2(uint64_t n) { return floor_log2(n) + ((n & (n - 1)) ? 1u : 0u); }

// include/bitops.hpp:170-183: number of equal leading characters of two k-mers
// stored with l bits per character, first character in the high bits of the
// k*l-bit field.
template <typename T> inline unsigned kmer_lcp(T x, T y, unsigned k, unsigned l) {
    if (x == y) return k;
    unsigned lz = lead_zeros<T>((T)(x ^ y));
    unsigned inside = lz - (unsigned)(sizeof(T) * 8 - k * l);
    return inside / l;
}

// ---------------------------------------------------------------- alphabet
// include/alphabet.hpp:49-59 (histogram), :147-164 (sigma, bits_per_char, codes 1..sigma
// in ascending unsigned byte order, 0 kept for the end marker).
struct Alpha {
    uint16_t code[256];
    unsigned sigma;
    unsigned bits;
};
Alpha make_alpha(const uint8_t* s, uint64_t n) {
    uint64_t hist[256];
    std::memset(hist, 0, sizeof(hist));
    for (uint64_t i = 0; i < n; ++i) hist[s[i]]++;
    Alpha a;
    uint16_t next = 1;
    for (int c = 0; c < 256; ++c) {
        if (hist[c]) a.code[c] = next++; else a.code[c] = 0;
    }
    a.sigma = next - 1u;
    a.bits = ceil_log2((uint64_t)a.sigma + 1u);
    return a;
}

// include/alphabet.hpp:254-262 + include/kmer.hpp:26-40 evaluated at p = 1.
unsigned pick_k(unsigned word_bits, unsigned l, uint64_t n, unsigned k) {
    unsigned max_k = word_bits / l;
    if (k == 0 || k > max_k) k = max_k;
    if ((uint64_t)k >= n) {
        k = (unsigned)n;
        if (k > 1) k--;           // comm.size() == 1 branch
    }
    return k;
}

// include/kmer.hpp:119-177 at p = 1 (last_kmer = 0): B[i] packs the codes of
// s[i..i+k), zero-filled past the end, first character most significant.
// k == 1: the reference copies raw characters (kmer.hpp:218-221, SURVEY D-1);
// this restatement always stores codes, which orders identically.
template <typename T>
void make_kmers(const uint8_t* s, uint64_t n, unsigned k, const Alpha& a, T* B) {
    const unsigned l = a.bits;
    T mask = (k * l >= sizeof(T) * 8) ? ~(T)0 : (T)(((T)1 << (k * l)) - 1);
    T cur = 0;
    // window holds the k-1 chars before position i+k-1
    for (unsigned j = 0; j + 1 < k; ++j) {
        cur = (T)(cur << l);
        if (j < n) cur |= (T)a.code[s[j]];
    }
    for (uint64_t i = 0; i < n; ++i) {
        uint64_t nxt = i + (k - 1);
        cur = (T)(cur << l);
        if (nxt < n) cur |= (T)a.code[s[nxt]];
        cur &= mask;
        B[i] = cur;
    }
}

// include/shifting.hpp:33-122 at p = 1: B2[i] = B[i+h], 0 past the end.
template <typename T>
void shift_left(const std::vector<T>& B, uint64_t h, std::vector<T>& B2) {
    const uint64_t n = B.size();
    B2.assign(n, 0);
    #pragma omp parallel for schedule(static)
    for (uint64_t i = 0; i < (h < n ? n - h : 0); ++i) B2[i] = B[i + h];
}

// include/idxsort.hpp:10-14, :23-83: (v1, v2, idx) tuples sorted by (v1, v2);
// mxx::sort degenerates to a local comparison sort at p = 1.
template <typename T> struct Tup { T a, b, idx; };
template <typename T>
void pair_sort(std::vector<T>& B1, std::vector<T>& B2, std::vector<T>& SA) {
    const uint64_t n = B1.size();
    std::vector<Tup<T> > t(n);
    #pragma omp parallel for schedule(static)
    for (uint64_t i = 0; i < n; ++i) { t[i].a = B1[i]; t[i].b = B2[i]; t[i].idx = (T)i; }
    std::sort(t.begin(), t.end(), [](const Tup<T>& x, const Tup<T>& y) {
        return x.a < y.a || (x.a == y.a && x.b < y.b);
    });
    SA.resize(n);
    #pragma omp parallel for schedule(static)
    for (uint64_t i = 0; i < n; ++i) { B1[i] = t[i].a; B2[i] = t[i].b; SA[i] = t[i].idx; }
}
// idxsort_vectors<T, T, true> (idxsort.hpp:23-83 with _STABLE): equal pairs keep their input order
template <typename T>
void pair_sort_stable(std::vector<T>& B1, std::vector<T>& B2, std::vector<T>& SA) {
    const uint64_t n = B1.size();
    std::vector<Tup<T> > t(n);
    #pragma omp parallel for schedule(static)
    for (uint64_t i = 0; i < n; ++i) { t[i].a = B1[i]; t[i].b = B2[i]; t[i].idx = (T)i; }
    std::stable_sort(t.begin(), t.end(), [](const Tup<T>& x, const Tup<T>& y) {
        return x.a < y.a || (x.a == y.a && x.b < y.b);
    });
    SA.resize(n);
    #pragma omp parallel for schedule(static)
    for (uint64_t i = 0; i < n; ++i) { B1[i] = t[i].a; B2[i] = t[i].b; SA[i] = t[i].idx; }
}

// include/bucketing.hpp:57-123 + :21-53 at p = 1.  In sorted order, a position
// whose (B1,B2) differs from its predecessor becomes a bucket head with id
// i+1 (1-based), the others 0; count buckets/elements still unresolved, then
// inclusive prefix-max fills the zeros.
// gsa_mask: bucketing.hpp:130-143 -- rebucket_gsa (mask = all ones: equal pairs stay together only
// if B2 != 0) and rebucket_gsa_kmers (mask = last character of the k-mer: only if the 2k-mer holds
// no end of string); 0 = the plain rule.
template <typename T>
void rebucket_pairs(std::vector<T>& B1, const std::vector<T>& B2, uint64_t& unf_b, uint64_t& unf_e, T gsa_mask = 0) {
    const uint64_t n = B1.size();
    std::vector<uint8_t> head(n);
    head[0] = 1;
#pragma omp parallel for schedule(static)
    for (uint64_t i = 1; i < n; ++i) {
        bool same = B1[i] == B1[i - 1] && B2[i] == B2[i - 1];
        if (gsa_mask && (B2[i - 1] & gsa_mask) == 0) same = false;
        head[i] = !same;
    }
    #pragma omp parallel for schedule(static)
    for (uint64_t i = 0; i < n; ++i) B1[i] = head[i] ? (T)(i + 1) : (T)0;
    unf_b = 0; unf_e = 0;
    for (uint64_t i = 1; i < n; ++i) {
        if (B1[i - 1] > 0 && B1[i] == 0) { ++unf_b; ++unf_e; }
        if (B1[i] == 0) ++unf_e;
    }
    T run = 0;
    for (uint64_t i = 0; i < n; ++i) { if (B1[i] == 0) B1[i] = run; else run = B1[i]; }
}

// include/bulk_permute.hpp:14-73 at p = 1: out[idx[i]] = val[i].
template <typename T>
void permute_to_isa(std::vector<T>& val, const std::vector<T>& idx) {
    std::vector<T> out(val.size());
    #pragma omp parallel for schedule(static)
    for (uint64_t i = 0; i < val.size(); ++i) out[idx[i]] = val[i];
    val.swap(out);
}

// Range-minimum over the LCP array returning the position of the LEFTMOST minimum
// (include/rmq.hpp:37-339; the value feeds LCP, suffix_array.hpp:1504, the position picks
// the left-branching character, par_rmq.hpp:426-431).  Blocks of 64 + sparse table of
// leftmost-minimum positions over the blocks.
template <typename T> struct RangeMin {
    const T* v; uint64_t n; uint64_t nb;
    std::vector<std::vector<uint64_t> > tab;
    uint64_t better(uint64_t a, uint64_t b) const { return v[b] < v[a] ? b : a; }   // a lies left of b
    RangeMin(const T* v_, uint64_t n_) : v(v_), n(n_) {
        nb = (n + 63) / 64;
        tab.emplace_back(nb);
        for (uint64_t b = 0; b < nb; ++b) {
            uint64_t m = b * 64;
            uint64_t e = std::min<uint64_t>(n, b * 64 + 64);
            for (uint64_t i = b * 64 + 1; i < e; ++i) m = better(m, i);
            tab[0][b] = m;
        }
        // tab[j][b] = leftmost minimum over blocks [b, b + 2^j)
        for (uint64_t span = 2; span <= nb; span <<= 1) {
            const std::vector<uint64_t>& p = tab.back();
            std::vector<uint64_t> c(nb - span + 1);
            for (uint64_t b = 0; b + span <= nb; ++b) c[b] = better(p[b], p[b + span / 2]);
            tab.push_back(std::move(c));
        }
    }
    uint64_t argmin(uint64_t l, uint64_t r) const {   // leftmost minimum of [l, r), l < r
        uint64_t bl = (l + 63) / 64, br = r / 64;
        uint64_t m = l;
        if (bl >= br) {
            for (uint64_t i = l + 1; i < r; ++i) m = better(m, i);
            return m;
        }
        for (uint64_t i = l + 1; i < bl * 64; ++i) m = better(m, i);
        uint64_t len = br - bl;
        unsigned lg = floor_log2(len);
        if (l < bl * 64) m = better(m, tab[lg][bl]); else m = tab[lg][bl];
        m = better(m, tab[lg][br - (1ull << lg)]);
        for (uint64_t i = br * 64; i < r; ++i) m = better(m, i);
        return m;
    }
    T query(uint64_t l, uint64_t r) const { return v[argmin(l, r)]; }
};

struct Trace {           // one line per refinement round
    uint64_t h;          // prefix length the sort of this round was keyed on / 2
    uint64_t unf_b;      // buckets still holding > 1 suffix afterwards
    uint64_t unf_e;      // suffixes in such buckets
    uint32_t phase;      // 0 = doubling loop (suffix_array.hpp:381-450), 1 = bucket chasing
};

template <typename T>
struct Engine {
    uint64_t n;
    std::vector<T> SA, B, LCP;
    std::vector<uint8_t> Lc;   // left-branching characters (suffix_array.hpp:211-212), want_lc only
    std::vector<Trace> trace;
    Alpha alpha;
    unsigned k;
    bool want_lcp, want_lc;
    Engine() : want_lcp(false), want_lc(false) {}

    // include/kmer.hpp:66-69 + alphabet.hpp:166-171, :281-283: character i of a packed k-mer,
    // decoded through the inverse mapping whose entry 0 is '\0'.
    uint8_t kmer_char(T kmer, unsigned i) const {
        const unsigned l = alpha.bits;
        unsigned code = (unsigned)((kmer >> ((k - 1 - i) * l)) & (((T)1 << l) - 1));
        if (code == 0) return 0;
        for (int c = 0; c < 256; ++c) if (alpha.code[c] == code) return (uint8_t)c;
        return 0;
    }

    // include/suffix_array.hpp:1353-1396 (both branches): LCP from the packed
    // 2k-mers at each bucket boundary of the first sort; sentinel n elsewhere.  With
    // _CONSTRUCT_LC the character of the LEFT k-mer at the mismatch is decoded (:1365-1383).
    void lcp_from_kmers(const std::vector<T>& B1, const std::vector<T>& B2) {
        LCP.assign(n, (T)n);
        LCP[0] = 0;
        if (want_lc) Lc.assign(n, 0);
        const unsigned l = alpha.bits;
#pragma omp parallel for schedule(static)
        for (uint64_t i = 1; i < n; ++i) {
            if (B1[i - 1] != B1[i] || B2[i - 1] != B2[i]) {
                unsigned v = kmer_lcp<T>(B1[i - 1], B1[i], k, l);
                if (v == k) {
                    unsigned v2 = kmer_lcp<T>(B2[i - 1], B2[i], k, l);
                    v += v2;
                    if (want_lc) Lc[i] = kmer_char(B2[i - 1], v2);
                } else if (want_lc) {
                    Lc[i] = kmer_char(B1[i - 1], v);
                }
                LCP[i] = (T)v;
            }
        }
    }

    // ---- generalized suffix array: strings s_0 .. s_{m-1} back to back without separators
    // (stringset.hpp:33-81), str_end[i] = end offset of the string holding position i.
    std::vector<uint64_t> str_end;

    // include/kmer.hpp:269-355 at p = 1: k-mers never run past the end of their string
    void make_kmers_ss(const uint8_t* s, const uint64_t* off, uint64_t m) {
        const unsigned l = alpha.bits;
        for (uint64_t t = 0; t < m; ++t) {
            for (uint64_t i = off[t]; i < off[t + 1]; ++i) {
                T w = 0;
                for (unsigned j = 0; j < k; ++j) {
                    w = (T)(w << l);
                    if (i + j < off[t + 1]) w |= (T)alpha.code[s[i + j]];
                }
                B[i] = w;
            }
        }
    }
    // include/shifting.hpp:374-418 at p = 1 (only inner sequences): B2[i] = B[i+h] inside the string
    void shift_ss(uint64_t h, std::vector<T>& B2) const {
        B2.assign(n, 0);
        for (uint64_t i = 0; i < n; ++i) if (i + h < str_end[i]) B2[i] = B[i + h];
    }
    // include/suffix_array.hpp:1404-1442 (initial_kmer_lcp_gsa)
    void lcp_from_kmers_gsa(const std::vector<T>& B1, const std::vector<T>& B2) {
        LCP.assign(n, (T)n);
        LCP[0] = 0;
        const unsigned l = alpha.bits;
        for (uint64_t i = 1; i < n; ++i) {
            const T l1 = B1[i - 1], l2 = B2[i - 1], r1 = B1[i], r2 = B2[i];
            if (l1 != r1) { LCP[i] = (T)kmer_lcp<T>(l1, r1, k, l); continue; }
            unsigned v = k - trail_zeros<T>(l1) / l;
            if (v == k) {
                if (l2 != r2) { v += kmer_lcp<T>(l2, r2, k, l); LCP[i] = (T)v; }
                else {
                    if (l2 != 0) v += k - trail_zeros<T>(l2) / l;
                    if (v < 2 * k) LCP[i] = (T)v;
                }
            } else LCP[i] = (T)v;
        }
    }
    // include/suffix_array.hpp:267-363 (construct_ss): one k-mer round, then always the
    // bucket-chasing tail (:333 "else if (true)"), whose B2 fetch stops at string ends (:998-1029).
    int construct_ss(const uint8_t* s, uint64_t n_, const uint64_t* off, uint64_t m, unsigned k_req, bool lcp) {
        n = n_; want_lcp = lcp; want_lc = false; trace.clear(); Lc.clear();
        if (n == 0 || m == 0 || off[0] != 0 || off[m] != n) return 1;
        for (uint64_t t = 0; t < m; ++t) if (off[t + 1] <= off[t]) return 1;
        alpha = make_alpha(s, n);
        k = pick_k((unsigned)sizeof(T) * 8, alpha.bits, n, k_req);
        str_end.assign(n, 0);
        for (uint64_t t = 0; t < m; ++t) for (uint64_t i = off[t]; i < off[t + 1]; ++i) str_end[i] = off[t + 1];
        B.assign(n, 0); SA.assign(n, 0); LCP.clear();
        if (lcp) LCP.assign(n, (T)n);
        if (n == 1) { SA[0] = 0; B[0] = 0; if (lcp) LCP[0] = 0; return 0; }
        make_kmers_ss(s, off, m);
        std::vector<T> Bsa;
        uint64_t unf_b = 1, unf_e = n;
        const uint64_t h = k;
        if (h < n) {
            std::vector<T> B2;
            shift_ss(h, B2);
            pair_sort_stable(B, B2, SA);
            if (lcp) lcp_from_kmers_gsa(B, B2);
            const T last_char = (T)(((T)1 << alpha.bits) - 1);
            rebucket_pairs(B, B2, unf_b, unf_e, last_char);
            Trace t = {h, unf_b, unf_e, 0};
            trace.push_back(t);
            if (!((h << 1) >= n || unf_b == 0)) Bsa = B;
            permute_to_isa(B, SA);
        } else return 2;
        if (unf_b > 0 && !Bsa.empty()) chase(Bsa, 2 * h);
        for (uint64_t i = 0; i < n; ++i) B[i] -= 1;
        return 0;
    }

    // include/par_rmq.hpp:199-332 (bulk_rmq_v2) / :334-481 (bulk_rmq_Lc) at p = 1, then the
    // update loops suffix_array.hpp:1485-1505 and :1221-1230: value h + min, and the
    // left-branching character stored at the leftmost minimum.
    void answer_ranges(uint64_t h, const std::vector<std::pair<uint64_t, uint64_t> >& q,
                       const std::vector<uint64_t>& where) {
        if (q.empty()) return;
        RangeMin<T> rm(LCP.data(), n);
        std::vector<uint64_t> pos(q.size());
        for (size_t j = 0; j < q.size(); ++j) pos[j] = rm.argmin(q[j].first, q[j].second);
        std::vector<T> ans(q.size());
        std::vector<uint8_t> ch(q.size(), 0);
        for (size_t j = 0; j < q.size(); ++j) { ans[j] = LCP[pos[j]]; if (want_lc) ch[j] = Lc[pos[j]]; }
        for (size_t j = 0; j < q.size(); ++j) {
            LCP[where[j]] = (T)(h + ans[j]);
            if (want_lc) Lc[where[j]] = ch[j];
        }
    }

    // include/suffix_array.hpp:1444-1508 + par_rmq.hpp:199-332 at p = 1.
    void lcp_from_ranges(uint64_t h, const std::vector<T>& B1, const std::vector<T>& B2) {
        std::vector<std::pair<uint64_t, uint64_t> > q;
        std::vector<uint64_t> where;
        for (uint64_t i = 1; i < n; ++i) {
            if (B1[i - 1] != B1[i]) continue;
            T x = B2[i - 1], y = B2[i];
            if (x == 0 || y == 0) { if (LCP[i] == (T)n) LCP[i] = (T)h; }
            else if (x != y) { q.emplace_back(std::min(x, y), std::max(x, y)); where.push_back(i); }
        }
        answer_ranges(h, q, where);
    }

    // include/suffix_array.hpp:925-965 at p = 1: positions (SA order) whose
    // bucket id differs from own index+1, plus heads followed by a member.
    void collect_active(const std::vector<T>& Bsa, std::vector<uint64_t>& act, bool first,
                        uint64_t& unresolved, uint64_t& unfinished) {
        std::vector<uint64_t> next;
        unresolved = unfinished = 0;
        uint64_t cnt = first ? n : act.size();
        for (uint64_t g = 0; g < cnt; ++g) {
            uint64_t j = first ? g : act[g];
            if (Bsa[j] != (T)(j + 1)) { next.push_back(j); ++unresolved; }
            else if (j + 1 < n && Bsa[j + 1] == (T)(j + 1)) { next.push_back(j); ++unresolved; ++unfinished; }
        }
        act.swap(next);
    }

    // include/suffix_array.hpp:1032-1285 at p = 1 (every bucket is an inner
    // bucket; no split buckets, no sub-communicators).  Bsa: bucket ids in SA
    // order; B: bucket ids in text order (becomes ISA+1).
    void chase(std::vector<T>& Bsa, uint64_t h0) {
        std::vector<uint64_t> act;
        uint64_t unres, unf;
        collect_active(Bsa, act, true, unres, unf);
        for (uint64_t h = h0; h < n; h <<= 1) {
            if (act.empty()) break;
            // sparse_get_b2: suffix_array.hpp:972-996
            std::vector<T> b2(act.size(), 0);
            for (size_t a = 0; a < act.size(); ++a) {
                uint64_t p = (uint64_t)SA[act[a]] + h;
                // suffix_array.hpp:972-996; with a string set the suffix h further must start
                // inside the same string (:998-1029)
                if (p < (str_end.empty() ? n : str_end[SA[act[a]]])) b2[a] = B[p];
            }
            std::vector<std::pair<uint64_t, uint64_t> > q;
            std::vector<uint64_t> where;
            size_t ai = 0;
            while (ai < act.size()) {
                const uint64_t beg = act[ai];            // bucket head position == id-1
                size_t a_beg = ai;
                uint64_t idx = act[ai];
                while (ai < act.size() && (uint64_t)Bsa[idx] - 1 == beg) { ++ai; ++idx; }
                size_t cnt = ai - a_beg;
                std::vector<size_t> ord(cnt);
                std::iota(ord.begin(), ord.end(), a_beg);
                // suffix_array.hpp:1115-1117
                std::sort(ord.begin(), ord.end(), [&](size_t x, size_t y) {
                    return b2[x] < b2[y] || (b2[x] == 0 && b2[y] == 0 && SA[act[x]] < SA[act[y]]);
                });
                std::vector<T> sa_copy(SA.begin() + beg, SA.begin() + beg + cnt);
                T cur = (T)(beg + 1);
                uint64_t out = beg;
                for (size_t t = 0; t < cnt; ++t) {
                    if (t > 0) {
                        T pb = b2[ord[t - 1]], cb = b2[ord[t]];
                        if (pb != cb || cb == 0) cur = (T)(out + 1);
                        if (want_lcp) {
                            if (pb == 0 || cb == 0) { if (LCP[out] == (T)n) LCP[out] = (T)h; }
                            else if (pb != cb) { q.emplace_back(std::min(pb, cb), std::max(pb, cb)); where.push_back(out); }
                        }
                    }
                    SA[out] = sa_copy[act[ord[t]] - beg];
                    Bsa[out] = cur;
                    ++out;
                }
            }
            if (want_lcp) answer_ranges(h, q, where);
            // suffix_array.hpp:1263-1277: push new ids to text order
            for (size_t a = 0; a < act.size(); ++a) B[SA[act[a]]] = Bsa[act[a]];
            collect_active(Bsa, act, false, unres, unf);
            Trace t = {h, unf, unres, 1};
            trace.push_back(t);
        }
    }

    // include/suffix_array.hpp:469-486 then :365-466.
    int construct(const uint8_t* s, uint64_t n_, bool fast, unsigned k_req, bool lcp, bool lc = false) {
        n = n_; want_lcp = lcp; want_lc = lcp && lc; trace.clear(); Lc.clear(); str_end.clear();
        if (want_lc) Lc.assign(n, 0);
        if (n == 0) return 1;
        alpha = make_alpha(s, n);
        k = pick_k((unsigned)sizeof(T) * 8, alpha.bits, n, k_req);
        B.assign(n, 0); SA.assign(n, 0); LCP.clear();
        if (n == 1) { SA[0] = 0; B[0] = 0; if (lcp) LCP.assign(1, 0); return 0; }
        make_kmers<T>(s, n, k, alpha, B.data());
        std::vector<T> Bsa;
        uint64_t unf_b = 1, unf_e = n, h;
        bool did_round = false;
        for (h = k; h < n; h <<= 1) {
            did_round = true;
            std::vector<T> B2;
            shift_left(B, h, B2);
            pair_sort(B, B2, SA);
            if (lcp) { if (h == k) lcp_from_kmers(B, B2); else lcp_from_ranges(h, B, B2); }
            rebucket_pairs(B, B2, unf_b, unf_e);
            Trace t = {h, unf_b, unf_e, 0};
            trace.push_back(t);
            if (fast && unf_e < n / 10) {
                Bsa = B;
                permute_to_isa(B, SA);
                break;
            }
            permute_to_isa(B, SA);
            if (unf_b == 0) break;
        }
        if (!did_round) return 2;
        if (unf_b > 0 && !Bsa.empty()) chase(Bsa, 2 * h);
        #pragma omp parallel for schedule(static)
        for (uint64_t i = 0; i < n; ++i) B[i] -= 1;      // suffix_array.hpp:460-464
        return 0;
    }
};

// include/lcp.hpp:46-77 (Kasai et al.), first entry 0.
template <typename T>
void kasai(const uint8_t* s, uint64_t n, const T* SA, const T* ISA, T* LCP) {
    if (n == 0) return;
    LCP[0] = 0;
    uint64_t h = 0;
    for (uint64_t i = 0; i < n; ++i) {
        uint64_t r = ISA[i];
        if (r == 0) { h = 0; continue; }
        uint64_t j = SA[r - 1];
        if (h > 0) --h;
        while (i + h < n && j + h < n && s[i + h] == s[j + h]) ++h;
        LCP[r] = (T)h;
    }
}

// include/check_suffix_array.hpp:56-88: SA/ISA consistency and suffix order
// through first character + rank of the suffix one to the right.
template <typename T>
int verify_sa(const uint8_t* s, uint64_t n, const T* SA, const T* ISA) {
    for (uint64_t i = 0; i < n; ++i) {
        if ((uint64_t)SA[i] >= n) return 1;
        if ((uint64_t)ISA[SA[i]] != i) return 2;
    }
    for (uint64_t i = 1; i < n; ++i) {
        uint64_t a = SA[i - 1], b = SA[i];
        if (s[a] > s[b]) return 3;
        if (s[a] == s[b]) {
            if (a + 1 == n) continue;             // shorter suffix first: fine
            if (b + 1 == n) return 4;
            if (ISA[a + 1] > ISA[b + 1]) return 5;
        }
    }
    return 0;
}

// Independent suffix sorter (direct suffix comparison); quadratic on repeats,
// used on small inputs only to cross-check the restatement.
template <typename T>
void naive_sa(const uint8_t* s, uint64_t n, T* SA) {
    for (uint64_t i = 0; i < n; ++i) SA[i] = (T)i;
    std::sort(SA, SA + n, [&](T a, T b) {
        uint64_t la = n - a, lb = n - b, m = std::min(la, lb);
        int c = std::memcmp(s + a, s + b, m);
        if (c != 0) return c < 0;
        return la < lb;
    });
}

// include/ansv.hpp:48-65 (ansv_sequential), nearest strictly smaller value.
template <typename T>
void ansv_seq(const T* in, uint64_t n, int left, uint64_t nonsv, uint64_t* out) {
    std::vector<uint64_t> st;
    for (uint64_t t = 0; t < n; ++t) {
        uint64_t i = left ? n - 1 - t : t;
        while (!st.empty() && in[i] < in[st.back()]) { out[st.back()] = i; st.pop_back(); }
        st.push_back(i);
    }
    for (uint64_t x : st) out[x] = nonsv;
}

// Result contract of ansv<T,left_type,right_type,global_indexing> (ansv.hpp:2042-2045,
// tie rules ansv_common.hpp:20-22, property checker test/test_ansv.cpp:35-135),
// restated by definition:
//   type 0 nearest_sm : nearest j with in[j] <  in[i]
//   type 1 nearest_eq : nearest j with in[j] <= in[i]
//   type 2 furthest_eq: let j = nearest_eq(i); walk on through equal values
//                       (in == in[j]) as long as nothing smaller lies between.
template <typename T>
void ansv_typed(const T* in, uint64_t n, int left, int type, uint64_t nonsv, uint64_t* out) {
    if (type == 0) { ansv_seq<T>(in, n, left, nonsv, out); return; }
    // stack of indices whose values are non-decreasing towards the top; run[k] = the lowest stack position that holds the same value
    // as position k with only equal values between (the walk "on through equal values" of furthest_eq, remembered instead of repeated:
    // an array whose minimum recurs without anything smaller between keeps every occurrence on the stack, and walking down through
    // them for every element is quadratic -- the many-tile ANSV tests of round 6 use such arrays)
    std::vector<uint64_t> st;
    std::vector<uint64_t> run;
    for (uint64_t t = 0; t < n; ++t) {
        uint64_t i = left ? t : n - 1 - t;
        // pop strictly larger elements
        while (!st.empty() && in[st.back()] > in[i]) { st.pop_back(); run.pop_back(); }
        // st.back() (if any) has value <= in[i]
        if (type == 1) out[i] = st.empty() ? nonsv : st.back();
        else out[i] = st.empty() ? nonsv : st[run.back()];
        run.push_back(!st.empty() && in[st.back()] == in[i] ? run.back() : (uint64_t)st.size());
        st.push_back(i);
    }
}

// include/suffix_tree.hpp:43-223 (for_each_parent) and :440-499 (construct_suffix_tree) at p = 1:
// one table row of sigma+1 cells per LCP index (= internal node), cell c holding the child
// reached through the character with alphabet code c (0 = end of text); leaves are n + i.
template <typename T>
void suffix_tree_nodes(const uint8_t* s, uint64_t n, const T* SA, const T* LCP, uint64_t* nodes /* n*(sigma+1), zeroed */,
                       uint32_t* sigma_out) {
    Alpha a = make_alpha(s, n);
    const uint64_t row = (uint64_t)a.sigma + 1;
    *sigma_out = a.sigma;
    const uint64_t NONE = ~0ull;
    std::vector<uint64_t> L(n), R(n);
    ansv_typed<T>(LCP, n, 1, 2, NONE, L.data());     // left: furthest_eq  (suffix_tree.hpp:62)
    ansv_typed<T>(LCP, n, 0, 0, NONE, R.data());     // right: nearest_sm
    auto add = [&](uint64_t parent, uint64_t gidx, uint64_t sa_val, uint64_t lcp_val) {
        const uint64_t ci = sa_val + lcp_val;                       // suffix_tree.hpp:461-468
        const uint64_t c = ci < n ? a.code[s[ci]] : 0;
        nodes[parent * row + c] = gidx;
    };
    // leaves (suffix_tree.hpp:72-143)
    for (uint64_t i = 0; i < n; ++i) {
        uint64_t parent, lcp_val;
        if (i == 0) {
            lcp_val = n > 1 ? (uint64_t)LCP[1] : 0;
            parent = lcp_val > 0 ? 1 : 0;
        } else if (i == n - 1 || LCP[i] >= LCP[i + 1]) {
            const uint64_t nsv = L[i];
            lcp_val = nsv != NONE ? (uint64_t)LCP[nsv] : 0;
            if (nsv != NONE && lcp_val == (uint64_t)LCP[i]) parent = nsv;
            else { parent = i; lcp_val = LCP[i]; }
        } else {
            parent = i + 1; lcp_val = LCP[i + 1];
        }
        add(parent, n + i, SA[i], lcp_val);
    }
    // internal nodes (suffix_tree.hpp:146-222)
    for (uint64_t i = 1; i < n; ++i) {
        if (LCP[i] == 0) continue;
        uint64_t parent, lcp_val;
        const uint64_t ln = L[i], rn = R[i];
        const uint64_t lv = LCP[ln];                                // exists: LCP[0] = 0
        if (rn == NONE) {
            if (lv == (uint64_t)LCP[i]) continue;
            parent = ln; lcp_val = lv;
        } else {
            const uint64_t rv = LCP[rn];
            if (lv >= rv) { if (lv == (uint64_t)LCP[i]) continue; parent = ln; lcp_val = lv; }
            else { parent = rn; lcp_val = rv; }
        }
        add(parent, i, SA[i], lcp_val);
    }
}

} // namespace

// ------------------------------------------------------------------ C ABI
extern "C" {

struct psac_ref_trace { uint64_t h, unfinished_buckets, unfinished_elements; uint32_t phase; uint32_t pad; };

#define DEFINE_FOR(T, SUF)                                                                         \
    int psac_ref_construct_##SUF(const uint8_t* text, uint64_t n, int fast, unsigned k, T* SA,     \
                                 T* ISA, T* LCP, psac_ref_trace* tr, uint32_t tr_cap,              \
                                 uint32_t* tr_len, uint32_t* k_used, uint32_t* bits_used) {        \
        Engine<T> e;                                                                               \
        int rc = e.construct(text, n, fast != 0, k, LCP != nullptr);                               \
        if (rc) return rc;                                                                         \
        std::memcpy(SA, e.SA.data(), n * sizeof(T));                                               \
        std::memcpy(ISA, e.B.data(), n * sizeof(T));                                               \
        if (LCP) std::memcpy(LCP, e.LCP.data(), n * sizeof(T));                                    \
        if (tr_len) {                                                                              \
            uint32_t m = (uint32_t)std::min<size_t>(e.trace.size(), tr_cap);                       \
            for (uint32_t i = 0; i < m && tr; ++i) {                                               \
                tr[i].h = e.trace[i].h; tr[i].unfinished_buckets = e.trace[i].unf_b;               \
                tr[i].unfinished_elements = e.trace[i].unf_e; tr[i].phase = e.trace[i].phase;      \
                tr[i].pad = 0;                                                                     \
            }                                                                                      \
            *tr_len = (uint32_t)e.trace.size();                                                    \
        }                                                                                          \
        if (k_used) *k_used = e.k;                                                                 \
        if (bits_used) *bits_used = e.alpha.bits;                                                  \
        return 0;                                                                                  \
    }                                                                                              \
    /* suffix_array<char, T, true, true>::construct (suffix_array.hpp:170, :469-486) */             \
    int psac_ref_construct_lc_##SUF(const uint8_t* text, uint64_t n, int fast, unsigned k, T* SA,  \
                                    T* ISA, T* LCP, uint8_t* Lc) {                                 \
        Engine<T> e;                                                                               \
        int rc = e.construct(text, n, fast != 0, k, true, true);                                   \
        if (rc) return rc;                                                                         \
        std::memcpy(SA, e.SA.data(), n * sizeof(T));                                               \
        std::memcpy(ISA, e.B.data(), n * sizeof(T));                                               \
        std::memcpy(LCP, e.LCP.data(), n * sizeof(T));                                             \
        std::memcpy(Lc, e.Lc.data(), n);                                                           \
        return 0;                                                                                  \
    }                                                                                              \
    /* suffix_array<char, T, LCP>::construct_ss (suffix_array.hpp:267-363): text = the strings back  \
     * to back, off[0..m] their offsets */                                                         \
    int psac_ref_construct_ss_##SUF(const uint8_t* text, uint64_t n, const uint64_t* off,          \
                                    uint64_t m, unsigned k, T* SA, T* ISA, T* LCP,                 \
                                    psac_ref_trace* tr, uint32_t tr_cap, uint32_t* tr_len) {       \
        Engine<T> e;                                                                               \
        int rc = e.construct_ss(text, n, off, m, k, LCP != nullptr);                               \
        if (rc) return rc;                                                                         \
        std::memcpy(SA, e.SA.data(), n * sizeof(T));                                               \
        std::memcpy(ISA, e.B.data(), n * sizeof(T));                                               \
        if (LCP) std::memcpy(LCP, e.LCP.data(), n * sizeof(T));                                    \
        if (tr_len) {                                                                              \
            uint32_t c = (uint32_t)std::min<size_t>(e.trace.size(), tr_cap);                       \
            for (uint32_t i = 0; i < c && tr; ++i) {                                               \
                tr[i].h = e.trace[i].h; tr[i].unfinished_buckets = e.trace[i].unf_b;               \
                tr[i].unfinished_elements = e.trace[i].unf_e; tr[i].phase = e.trace[i].phase;      \
                tr[i].pad = 0;                                                                     \
            }                                                                                      \
            *tr_len = (uint32_t)e.trace.size();                                                    \
        }                                                                                          \
        return 0;                                                                                  \
    }                                                                                              \
    void psac_ref_kmers_##SUF(const uint8_t* text, uint64_t n, unsigned k, T* out) {               \
        Alpha a = make_alpha(text, n);                                                             \
        make_kmers<T>(text, n, k, a, out);                                                         \
    }                                                                                              \
    unsigned psac_ref_lcp_bitwise_##SUF(T x, T y, unsigned k, unsigned l) {                        \
        return kmer_lcp<T>(x, y, k, l);                                                            \
    }                                                                                              \
    unsigned psac_ref_leading_zeros_##SUF(T x) { return lead_zeros<T>(x); }                        \
    unsigned psac_ref_trailing_zeros_##SUF(T x) { return trail_zeros<T>(x); }                      \
    void psac_ref_rebucket_##SUF(T* b1, const T* b2, uint64_t n, uint64_t* ub, uint64_t* ue) {     \
        std::vector<T> v1(b1, b1 + n), v2(b2, b2 + n);                                             \
        rebucket_pairs<T>(v1, v2, *ub, *ue);                                                       \
        std::memcpy(b1, v1.data(), n * sizeof(T));                                                 \
    }                                                                                              \
    void psac_ref_kasai_##SUF(const uint8_t* text, uint64_t n, const T* SA, const T* ISA, T* L) {  \
        kasai<T>(text, n, SA, ISA, L);                                                             \
    }                                                                                              \
    int psac_ref_check_sa_##SUF(const uint8_t* text, uint64_t n, const T* SA, const T* ISA) {      \
        return verify_sa<T>(text, n, SA, ISA);                                                     \
    }                                                                                              \
    void psac_ref_naive_sa_##SUF(const uint8_t* text, uint64_t n, T* SA) {                         \
        naive_sa<T>(text, n, SA);                                                                  \
    }                                                                                              \
    void psac_ref_ansv_seq_##SUF(const T* in, uint64_t n, int left, uint64_t nonsv,                \
                                 uint64_t* out) {                                                  \
        ansv_seq<T>(in, n, left, nonsv, out);                                                      \
    }                                                                                              \
    void psac_ref_ansv_##SUF(const T* in, uint64_t n, int left, int type, uint64_t nonsv,          \
                             uint64_t* out) {                                                      \
        ansv_typed<T>(in, n, left, type, nonsv, out);                                              \
    }                                                                                              \
    void psac_ref_suffix_tree_##SUF(const uint8_t* text, uint64_t n, const T* SA, const T* LCP,       \
                                    uint64_t* nodes, uint32_t* sigma) {                                \
        suffix_tree_nodes<T>(text, n, SA, LCP, nodes, sigma);                                          \
    }                                                                                                  \
    T psac_ref_range_min_##SUF(const T* v, uint64_t n, uint64_t l, uint64_t r) {                   \
        RangeMin<T> rm(v, n);                                                                      \
        return rm.query(l, r);                                                                     \
    }

DEFINE_FOR(uint32_t, u32)
DEFINE_FOR(uint64_t, u64)

void psac_ref_alphabet(const uint8_t* text, uint64_t n, uint16_t* code256, uint32_t* sigma,
                       uint32_t* bits) {
    Alpha a = make_alpha(text, n);
    std::memcpy(code256, a.code, sizeof(a.code));
    *sigma = a.sigma; *bits = a.bits;
}
unsigned psac_ref_optimal_k(unsigned word_bits, unsigned l, uint64_t n, unsigned k) {
    return pick_k(word_bits, l, n, k);
}
unsigned psac_ref_floorlog2(uint64_t x) { return floor_log2(x); }
unsigned psac_ref_ceillog2(uint64_t x) { return ceil_log2(x); }

// include/alphabet.hpp:32-45: srand(1337*seed); "ACGT"[rand() % 4] (glibc rand).
void psac_ref_rand_dna(uint64_t n, int seed, uint8_t* out) {
    static const char dna[4] = {'A', 'C', 'G', 'T'};
    srand(1337 * seed);
    for (uint64_t i = 0; i < n; ++i) out[i] = (uint8_t)dna[rand() % 4];
}

// FNV-1a-64 over values widened to uint64, little endian (SURVEY.md Appendix C checksums).
uint64_t psac_ref_fnv64_u64(const uint64_t* v, uint64_t n) {
    uint64_t h = 0xcbf29ce484222325ull;
    for (uint64_t i = 0; i < n; ++i) {
        uint64_t x = v[i];
        for (int b = 0; b < 8; ++b) { h ^= (x >> (8 * b)) & 0xff; h *= 0x100000001b3ull; }
    }
    return h;
}
uint64_t psac_ref_fnv64_u32(const uint32_t* v, uint64_t n) {
    uint64_t h = 0xcbf29ce484222325ull;
    for (uint64_t i = 0; i < n; ++i) {
        uint64_t x = v[i];
        for (int b = 0; b < 8; ++b) { h ^= (x >> (8 * b)) & 0xff; h *= 0x100000001b3ull; }
    }
    return h;
}

} // extern "C"
