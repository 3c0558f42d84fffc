// suffix_array.hpp -- C++11 host mirror of psac's suffix_array<> class over libpsacx.so.
//
// Same class name, template parameters, public fields and construct() signatures as
// /root/reference/include/suffix_array.hpp:170-228, :365-486, so that a caller such as
// src/psac.cpp:117-128 compiles against this header unchanged apart from the communicator
// type: the reference takes an mxx::comm (one MPI rank per text block); this engine takes a
// psacx::comm naming the HIP device(s): one device = one rank; several devices = that many ranks,
// one GPU each, all driven by this process (the text is block-decomposed over them exactly as
// src/psac.cpp:85-93 decomposes it over MPI ranks, the exchanges run over RCCL).  All compute happens in the
// HIP engine behind the C ABI of psacx.h; this header only moves data and re-throws errors
// (std::runtime_error, as suffix_array.hpp:226-227 does).
#ifndef PSACX_SUFFIX_ARRAY_HPP
#define PSACX_SUFFIX_ARRAY_HPP

#include <algorithm>
#include <cstddef>
#include <cstdint>
#include <fstream>
#include <limits>
#include <iostream>
#include <iterator>
#include <stdexcept>
#include <string>
#include <type_traits>
#include <vector>

#include "psacx.h"

namespace psacx {

// Stand-in for mxx::comm (Appendix A of SURVEY.md lists the surface psac uses).  One process holds every rank:
// rank() is 0, size() the number of GPUs (ranks) the communicator spans.
class comm {
public:
    explicit comm(int device = 0) : devices_(1, device) {}
    explicit comm(const std::vector<int>& devices) : devices_(devices.empty() ? std::vector<int>(1, 0) : devices) {}
    int rank() const { return 0; }
    int size() const { return (int)devices_.size(); }
    bool is_first() const { return true; }
    int device() const { return devices_[0]; }
    const std::vector<int>& devices() const { return devices_; }
    comm copy() const { return comm(devices_); }
private:
    std::vector<int> devices_;
};

// mxx::blk_dist at one rank (suffix_array.hpp:194, bulk_permute.hpp:23)
struct blk_dist {
    std::size_t n;
    blk_dist() : n(0) {}
    explicit blk_dist(std::size_t n_) : n(n_) {}
    std::size_t global_size() const { return n; }
    std::size_t local_size() const { return n; }
    std::size_t eprefix_size() const { return 0; }
    std::size_t iprefix_size() const { return n; }
    int rank_of(std::size_t) const { return 0; }
};

// characters compare by their unsigned value (alphabet.hpp:205-236)
template <typename char_t> struct unsigned_less {
    bool operator()(char_t a, char_t b) const { typedef typename std::make_unsigned<char_t>::type u; return (u)a < (u)b; }
};

// the part of alphabet<char> callers read (alphabet.hpp:147-164, :224-262, :296-300)
template <typename char_t> class alphabet {
public:
    alphabet() : m_sigma(0), m_bits(0) {}
    unsigned int sigma() const { return m_sigma; }
    unsigned int size() const { return m_sigma; }
    unsigned int bits_per_char() const { return m_bits; }
    template <typename word_type> unsigned int chars_per_word() const {
        unsigned int b = sizeof(word_type) * 8;
        if (std::is_signed<word_type>::value) --b;
        return b / m_bits;
    }
    const std::vector<char_t>& unique_chars() const { return m_chars; }
    void write(const std::string& filename) const {
        std::ofstream f(filename.c_str(), std::ios::binary);
        for (std::size_t i = 0; i < m_chars.size(); ++i) f.write(reinterpret_cast<const char*>(&m_chars[i]), sizeof(char_t));
    }
    // alphabet.hpp:205-236: the characters that occur, ascending by unsigned value
    template <typename Iterator>
    static alphabet from_sequence(Iterator begin, Iterator end) {
        typedef typename std::make_unsigned<char_t>::type uchar_t;
        std::vector<uchar_t> u;
        for (Iterator it = begin; it != end; ++it) u.push_back((uchar_t)*it);
        std::sort(u.begin(), u.end());
        u.erase(std::unique(u.begin(), u.end()), u.end());
        std::vector<char_t> chars;
        for (std::size_t i = 0; i < u.size(); ++i) chars.push_back((char_t)u[i]);
        alphabet a; a.set(chars); return a;
    }
    template <typename Iterator, typename Comm>
    static alphabet from_sequence(Iterator begin, Iterator end, const Comm&) { return from_sequence(begin, end); }
    static alphabet from_string(const std::basic_string<char_t>& str) { return from_sequence(str.begin(), str.end()); }
    template <typename Comm>
    static alphabet from_string(const std::basic_string<char_t>& str, const Comm&) { return from_sequence(str.begin(), str.end()); }
    template <typename StringSet, typename Comm>
    static alphabet from_stringset(const StringSet& ss, const Comm&) {
        std::basic_string<char_t> all;
        for (std::size_t s = 0; s < ss.sizes.size(); ++s) all.append(ss.str_begins[s], ss.str_begins[s] + ss.sizes[s]);
        return from_string(all);
    }
    // alphabet.hpp:303-311, :328-331: the used characters, ascending
    void read(const std::string& filename) {
        std::ifstream f(filename.c_str(), std::ios::binary);
        if (!f) throw std::runtime_error("cannot read " + filename);
        std::vector<char_t> chars; char_t c;
        while (f.read(reinterpret_cast<char*>(&c), sizeof(char_t))) chars.push_back(c);
        set(chars);
    }
    bool operator==(const alphabet& o) const { return m_chars == o.m_chars; }     // alphabet.hpp:286-288
    bool operator!=(const alphabet& o) const { return !(*this == o); }
    void set(const std::vector<char_t>& chars) {
        m_chars = chars; m_sigma = (unsigned int)chars.size();
        m_bits = 0; while ((1u << m_bits) < m_sigma + 1u) ++m_bits;
    }
private:
    std::vector<char_t> m_chars;
    unsigned int m_sigma, m_bits;
};

template <typename char_t> std::ostream& operator<<(std::ostream& os, const alphabet<char_t>& a) {
    os << "{sigma=" << a.sigma() << ", l=" << a.bits_per_char() << ", A=[";
    for (std::size_t i = 0; i < a.unique_chars().size(); ++i) os << (i ? ", " : "") << a.unique_chars()[i];
    return os << "]}";
}

inline void check(psacx_ctx* ctx, int rc) {
    if (rc == PSACX_OK) return;
    std::string msg = std::string("psacx: ") + psacx_strerror(rc);
    const char* d = ctx ? psacx_last_hip_error(ctx) : "";
    if (d && *d) msg += std::string(" [") + d + "]";
    throw std::runtime_error(msg);
}

} // namespace psacx

// simple_dstringset of /root/reference/include/stringset.hpp:33-151 on one rank: the strings of a
// flat buffer, cut at runs of the separator; empty strings do not exist.
class simple_dstringset {
public:
    bool first_split, last_split;              // never set on one rank (stringset.hpp:74-80)
    std::vector<const char*> str_begins;
    std::vector<std::size_t> sizes;
    std::size_t sum_sizes;

    template <typename Iterator>
    simple_dstringset(Iterator begin, Iterator end, const psacx::comm&, char sep = '$')
        : first_split(false), last_split(false), sum_sizes(0) {
        Iterator it = begin;
        while (it != end && *it == sep) ++it;
        while (it != end) {
            Iterator e = it;
            while (e != end && *e != sep) ++e;
            sizes.push_back((std::size_t)std::distance(it, e));
            sum_sizes += sizes.back();
            str_begins.push_back(&(*it));
            it = e;
            while (it != end && *it == sep) ++it;
        }
    }
};

// stringset.hpp:586-600: every string followed by the separator
inline std::string flatten_strings(const std::vector<std::string>& strs, char sep = '$') {
    std::string out;
    for (std::size_t i = 0; i < strs.size(); ++i) { out += strs[i]; out.push_back(sep); }
    return out;
}

#ifndef PSACX_INFO
#define PSACX_INFO(msg) { std::cerr << msg << std::endl; }
#endif

template <typename char_t, typename index_t = std::size_t, bool _CONSTRUCT_LCP = false, bool _CONSTRUCT_LC = false>
class suffix_array {
    static_assert(sizeof(index_t) == 4 || sizeof(index_t) == 8, "index_t must be a 32 or 64 bit unsigned integer");
    static_assert(!_CONSTRUCT_LC || _CONSTRUCT_LCP, "_CONSTRUCT_LC needs _CONSTRUCT_LCP (the reference fills Lc inside its LCP code)");
public:
    explicit suffix_array(const psacx::comm& _comm)
        : n(0), local_size(0), comm(_comm.copy()), p(_comm.size()), verbose(true), ctx_(nullptr), multi_(nullptr) {
        psacx::check(nullptr, psacx_create(&ctx_, comm.device(), nullptr));
        if (p > 1) {
            const int rc = psacx_multi_create(&multi_, p, comm.devices().data());
            if (rc != PSACX_OK) { psacx_destroy(ctx_); ctx_ = nullptr; psacx::check(nullptr, rc); }
        }
    }
    virtual ~suffix_array() { if (multi_) psacx_multi_destroy(multi_); if (ctx_) psacx_destroy(ctx_); }
    suffix_array(const suffix_array&) = delete;
    suffix_array& operator=(const suffix_array&) = delete;

    /// The global size of the input string and suffix array (suffix_array.hpp:180)
    std::size_t n;
    /// The local size (== n: this process holds the blocks of all p ranks, in rank order) (suffix_array.hpp:185)
    std::size_t local_size;
    psacx::comm comm;
    /// number of ranks = GPUs the construction is distributed over (suffix_array.hpp:191)
    int p;
    psacx::blk_dist part;
    using char_type = char_t;
    using alphabet_type = psacx::alphabet<char_t>;
    alphabet_type alpha;
    /// The suffix array, the inverse suffix array (0-based) and the LCP array
    /// (suffix_array.hpp:204-209); local_LCP stays empty unless _CONSTRUCT_LCP.
    std::vector<index_t> local_SA;
    std::vector<index_t> local_B;
    std::vector<index_t> local_LCP;
    /// left-branching characters Lc[i] = S[SA[i-1] + LCP[i]] (suffix_array.hpp:211-212), '\0' past
    /// the end; stays empty unless _CONSTRUCT_LC
    std::vector<char_t> local_Lc;
    bool verbose;                     // print the reference's stderr lines

    void init_size(std::size_t lsize) {      // suffix_array.hpp:217-228
        local_size = lsize; n = lsize; part = psacx::blk_dist(n);
    }

    // suffix_array.hpp:469-486
    template <typename Iterator>
    void construct(Iterator begin, Iterator end, bool fast_resolval = true, unsigned int k = 0) {
        construct_with(begin, end, fast_resolval, k, nullptr);
    }
    // suffix_array.hpp:365-366: the alphabet and k come from the caller (an alphabet that covers more characters than occur is
    // fine: SA, ISA and LCP do not depend on the coding, which only has to keep the order of the characters; a character
    // outside it is an error).  `alpha` is the caller's afterwards, as in the reference.
    template <typename Iterator>
    void construct(Iterator begin, Iterator end, bool fast_resolval, const alphabet_type& a, unsigned int k) {
        construct_with(begin, end, fast_resolval, k, &a);
    }

private:
    template <typename Iterator>
    void construct_with(Iterator begin, Iterator end, bool fast_resolval, unsigned int k, const alphabet_type* given) {
        init_size((std::size_t)std::distance(begin, end));
        if (n == 0) throw std::runtime_error("psacx: empty input");
        std::vector<uint8_t> bytes(n);
        std::vector<char_t> chars;
        const unsigned w = densify(begin, end, bytes, chars);
        if (given) {
            const std::vector<char_t>& have = given->unique_chars();
            for (std::size_t i = 0; i < chars.size(); ++i)
                if (!std::binary_search(have.begin(), have.end(), chars[i], psacx::unsigned_less<char_t>()))
                    throw std::runtime_error("psacx: the text holds a character that is not in the given alphabet");
            if (k == 0) throw std::runtime_error("psacx: construct with an explicit alphabet needs k > 0");
            const unsigned int kmax = given->template chars_per_word<index_t>();
            if (k > kmax) k = kmax;
        }
        local_SA.assign(n, 0); local_B.assign(n, 0);
        if (_CONSTRUCT_LCP) local_LCP.assign(n, 0); else local_LCP.clear();
        uint32_t flags = (_CONSTRUCT_LCP ? PSACX_LCP : 0u) | (fast_resolval ? 0u : PSACX_NO_FAST);
        std::vector<uint8_t> lc;
        if (_CONSTRUCT_LC) lc.assign(n, 0);
        int rc;
        if (w > 1) {
            // more than 256 distinct symbols: the engine builds the arrays of the text of w bytes per symbol, reduce_wide keeps the
            // suffixes that start on a symbol boundary
            if ((uint64_t)n * w > (uint64_t)std::numeric_limits<index_t>::max())
                throw std::runtime_error("psacx: the text of this many distinct symbols needs a wider index type");
            if (multi_ && !fast_resolval) throw std::runtime_error("psacx: fast_resolval = false needs a single-rank communicator");
            struct length_guard {                       // run() and run_multi() read the length from the object
                std::size_t& n; std::size_t n0;
                length_guard(std::size_t& n_, std::size_t bytes_) : n(n_), n0(n_) { n = bytes_; }
                ~length_guard() { n = n0; }
            };
            std::vector<index_t> sa((std::size_t)n * w), isa((std::size_t)n * w), lcp;
            if (_CONSTRUCT_LCP) lcp.assign((std::size_t)n * w, 0);
            {
                length_guard g(n, (std::size_t)n * w);
                rc = multi_ ? run_multi(bytes.data(), k * w, flags, sa.data(), isa.data(), _CONSTRUCT_LCP ? lcp.data() : nullptr, nullptr)
                            : run(bytes.data(), k * w, flags, sa.data(), isa.data(), _CONSTRUCT_LCP ? lcp.data() : nullptr, nullptr);
            }
            if (multi_) {
                if (rc != PSACX_OK) throw std::runtime_error(std::string("psacx: ") + (rc > -7 ? psacx_strerror(rc) : "RCCL failure") + " [" +
                                                             psacx_multi_last_error(multi_) + "]");
            } else psacx::check(ctx_, rc);
            reduce_wide(w, sa, lcp);
        } else if (multi_) {
            // p ranks, one GPU each: blocks of n / p characters (mxx::blk_dist), results gathered in rank order
            if (!fast_resolval) throw std::runtime_error("psacx: fast_resolval = false needs a single-rank communicator");
            rc = run_multi(bytes.data(), k, flags, local_SA.data(), local_B.data(), _CONSTRUCT_LCP ? local_LCP.data() : nullptr,
                           _CONSTRUCT_LC ? lc.data() : nullptr);
            if (rc != PSACX_OK) throw std::runtime_error(std::string("psacx: ") + (rc > -7 ? psacx_strerror(rc) : "RCCL failure") + " [" +
                                                         psacx_multi_last_error(multi_) + "]");
        } else {
            rc = run(bytes.data(), k, flags, local_SA.data(), local_B.data(), _CONSTRUCT_LCP ? local_LCP.data() : nullptr,
                     _CONSTRUCT_LC ? lc.data() : nullptr);
            psacx::check(ctx_, rc);
        }
        local_Lc.clear();
        if (_CONSTRUCT_LC) {
            // positions past the end carry '\0' (alphabet.hpp:168); they are exactly those with SA[i-1] + LCP[i] == n
            local_Lc.assign(n, (char_t)0);
            std::vector<char_t> sym;
            if (w > 1) sym.assign(begin, end);          // (wide symbols: the character itself, read from the caller's text)
            for (std::size_t i = 1; i < n; ++i) {
                const std::size_t at = (std::size_t)local_SA[i - 1] + (std::size_t)local_LCP[i];
                if (at >= n) continue;
                local_Lc[i] = w > 1 ? sym[at] : sizeof(char_t) == 1 ? (char_t)lc[i] : chars[lc[i]];
            }
        }
        psacx_stats st;
        if (multi_) psacx::check(nullptr, psacx_multi_get_stats(multi_, &st, nullptr, nullptr, nullptr));
        else psacx::check(ctx_, psacx_get_stats(ctx_, &st));
        if (given) alpha = *given; else alpha.set(chars);
        if (verbose) {
            PSACX_INFO("Alphabet: " << alpha);                       // suffix_array.hpp:481
            for (uint32_t r = 0; r < st.n_rounds; ++r)                // suffix_array.hpp:416
                PSACX_INFO("iteration " << st.rounds[r].h << ": unfinished buckets = " << st.rounds[r].unfinished_buckets
                           << ", unfinished elements = " << st.rounds[r].unfinished_elements);
        }
    }

public:
    // suffix_array.hpp:267-363: generalized suffix array of a string set.  local_SA counts positions
    // in the strings laid back to back without separators; equal suffixes come in that order.
    void construct_ss(simple_dstringset& ss, const alphabet_type& a) {
        static_assert(sizeof(char_t) == 1, "string sets hold bytes");
        static_assert(!_CONSTRUCT_LC, "left-branching characters are not defined for string sets");
        init_size(ss.sum_sizes);
        if (n == 0) throw std::runtime_error("psacx: empty input");
        std::vector<uint8_t> bytes; bytes.reserve(n);
        std::vector<uint64_t> off(1, 0);
        for (std::size_t s = 0; s < ss.sizes.size(); ++s) {
            bytes.insert(bytes.end(), reinterpret_cast<const uint8_t*>(ss.str_begins[s]),
                         reinterpret_cast<const uint8_t*>(ss.str_begins[s]) + ss.sizes[s]);
            off.push_back(bytes.size());
        }
        local_SA.assign(n, 0); local_B.assign(n, 0);
        if (_CONSTRUCT_LCP) local_LCP.assign(n, 0); else local_LCP.clear();
        const uint32_t flags = _CONSTRUCT_LCP ? PSACX_LCP : 0u;
        psacx_stats st;
        if (multi_) {
            // p > 1: the strings lie back to back in the block-distributed text (psacx_multi_construct_gsa_*)
            const int rc = run_gsa_multi(bytes.data(), off.data(), (uint64_t)ss.sizes.size(), flags, local_SA.data(), local_B.data(),
                                         _CONSTRUCT_LCP ? local_LCP.data() : nullptr);
            if (rc != PSACX_OK) throw std::runtime_error(std::string("psacx: ") + psacx_strerror(rc) + " [" + psacx_multi_last_error(multi_) + "]");
            psacx::check(nullptr, psacx_multi_get_stats(multi_, &st, nullptr, nullptr, nullptr));
        } else {
            psacx::check(ctx_, run_gsa(bytes.data(), off.data(), (uint64_t)ss.sizes.size(), flags, local_SA.data(), local_B.data(),
                                       _CONSTRUCT_LCP ? local_LCP.data() : nullptr));
            psacx::check(ctx_, psacx_get_stats(ctx_, &st));
        }
        alpha = a;
        if (verbose) {
            PSACX_INFO("Alphabet: " << alpha);                       // suffix_array.hpp:273-275
            for (uint32_t r = 0; r < st.n_rounds; ++r)                // suffix_array.hpp:317
                PSACX_INFO("iteration " << st.rounds[r].h << ": unfinished buckets = " << st.rounds[r].unfinished_buckets
                           << ", unfinished elements = " << st.rounds[r].unfinished_elements);
        }
    }

    // suffix_array.hpp:232-242: raw little-endian arrays, no header
    void write(const std::string& basename) const {
        dump(basename + ".sa", local_SA);
        if (_CONSTRUCT_LCP) dump(basename + ".lcp", local_LCP);
        if (_CONSTRUCT_LC) dump(basename + ".lc", local_Lc);
        alpha.write(basename + ".alpha");
    }
    // suffix_array.hpp:245-265
    void read(const std::string& basename) {
        slurp(basename + ".sa", local_SA);
        if (_CONSTRUCT_LCP) {
            slurp(basename + ".lcp", local_LCP);
            if (local_SA.size() != local_LCP.size()) throw std::runtime_error("SA and LCP have to have same size");
        }
        if (_CONSTRUCT_LC) {
            slurp(basename + ".lc", local_Lc);
            if (local_SA.size() != local_Lc.size()) throw std::runtime_error("SA and Lc have to have same size");
        }
        alpha.read(basename + ".alpha");
        init_size(local_SA.size());
    }

    psacx_ctx* context() { return ctx_; }
    psacx_multi* multi_context() { return multi_; }

private:
    psacx_ctx* ctx_;
    psacx_multi* multi_;

    // (lc != nullptr: also the left-branching characters, psacx_multi_construct_lc_*)
    int run_multi(const uint8_t* t, unsigned int k, uint32_t flags, uint32_t* sa, uint32_t* isa, uint32_t* lcp, uint8_t* lc) {
        return lc ? psacx_multi_construct_lc_u32(multi_, t, n, k, flags, sa, isa, lcp, lc) : psacx_multi_construct_u32(multi_, t, n, k, flags, sa, isa, lcp);
    }
    int run_multi(const uint8_t* t, unsigned int k, uint32_t flags, uint64_t* sa, uint64_t* isa, uint64_t* lcp, uint8_t* lc) {
        return lc ? psacx_multi_construct_lc_u64(multi_, t, n, k, flags, sa, isa, lcp, lc) : psacx_multi_construct_u64(multi_, t, n, k, flags, sa, isa, lcp);
    }
    template <typename U>
    typename std::enable_if<!std::is_same<U, uint32_t>::value && !std::is_same<U, uint64_t>::value, int>::type
    run_multi(const uint8_t* t, unsigned int k, uint32_t flags, U* sa, U* isa, U* lcp, uint8_t* lc) {
        typedef typename std::conditional<sizeof(U) == 4, uint32_t, uint64_t>::type W;
        return run_multi(t, k, flags, reinterpret_cast<W*>(sa), reinterpret_cast<W*>(isa), reinterpret_cast<W*>(lcp), lc);
    }

    int run(const uint8_t* t, unsigned int k, uint32_t flags, uint32_t* sa, uint32_t* isa, uint32_t* lcp, uint8_t* lc) {
        return lc ? psacx_construct_lc_u32(ctx_, t, n, k, flags, sa, isa, lcp, lc) : psacx_construct_u32(ctx_, t, n, k, flags, sa, isa, lcp);
    }
    int run(const uint8_t* t, unsigned int k, uint32_t flags, uint64_t* sa, uint64_t* isa, uint64_t* lcp, uint8_t* lc) {
        return lc ? psacx_construct_lc_u64(ctx_, t, n, k, flags, sa, isa, lcp, lc) : psacx_construct_u64(ctx_, t, n, k, flags, sa, isa, lcp);
    }
    int run_gsa(const uint8_t* t, const uint64_t* off, uint64_t m, uint32_t flags, uint32_t* sa, uint32_t* isa, uint32_t* lcp) {
        return psacx_construct_gsa_u32(ctx_, t, n, off, m, 0, flags, sa, isa, lcp);
    }
    int run_gsa(const uint8_t* t, const uint64_t* off, uint64_t m, uint32_t flags, uint64_t* sa, uint64_t* isa, uint64_t* lcp) {
        return psacx_construct_gsa_u64(ctx_, t, n, off, m, 0, flags, sa, isa, lcp);
    }
    int run_gsa_multi(const uint8_t* t, const uint64_t* off, uint64_t m, uint32_t flags, uint32_t* sa, uint32_t* isa, uint32_t* lcp) {
        return psacx_multi_construct_gsa_u32(multi_, t, n, off, m, 0, flags, sa, isa, lcp);
    }
    int run_gsa_multi(const uint8_t* t, const uint64_t* off, uint64_t m, uint32_t flags, uint64_t* sa, uint64_t* isa, uint64_t* lcp) {
        return psacx_multi_construct_gsa_u64(multi_, t, n, off, m, 0, flags, sa, isa, lcp);
    }
    template <typename U>
    typename std::enable_if<!std::is_same<U, uint32_t>::value && !std::is_same<U, uint64_t>::value, int>::type
    run_gsa_multi(const uint8_t* t, const uint64_t* off, uint64_t m, uint32_t flags, U* sa, U* isa, U* lcp) {
        typedef typename std::conditional<sizeof(U) == 4, uint32_t, uint64_t>::type W;
        return run_gsa_multi(t, off, m, flags, reinterpret_cast<W*>(sa), reinterpret_cast<W*>(isa), reinterpret_cast<W*>(lcp));
    }
    template <typename U>
    typename std::enable_if<!std::is_same<U, uint32_t>::value && !std::is_same<U, uint64_t>::value, int>::type
    run_gsa(const uint8_t* t, const uint64_t* off, uint64_t m, uint32_t flags, U* sa, U* isa, U* lcp) {
        typedef typename std::conditional<sizeof(U) == 4, uint32_t, uint64_t>::type W;
        return run_gsa(t, off, m, flags, reinterpret_cast<W*>(sa), reinterpret_cast<W*>(isa), reinterpret_cast<W*>(lcp));
    }
    template <typename U>
    typename std::enable_if<!std::is_same<U, uint32_t>::value && !std::is_same<U, uint64_t>::value, int>::type
    run(const uint8_t* t, unsigned int k, uint32_t flags, U* sa, U* isa, U* lcp, uint8_t* lc) {
        typedef typename std::conditional<sizeof(U) == 4, uint32_t, uint64_t>::type W;
        return run(t, k, flags, reinterpret_cast<W*>(sa), reinterpret_cast<W*>(isa), reinterpret_cast<W*>(lcp), lc);
    }

    // bytes pass through; wider symbols (int alphabets, test/test_psac.cpp:277-304) are ranked among the distinct symbols that occur,
    // which keeps their order.  At most 256 of them: one byte per symbol.  More (the reference's alphabet<int> is unbounded,
    // alphabet.hpp:205-236): the rank as w = 2, 3 or 4 bytes, most significant first -- fixed-width big-endian codes compare as the
    // symbols do, so the suffixes of the byte text that start on a symbol boundary stand in the order of the symbol text's suffixes
    // (reduce_wide below).  Returns w.
    template <typename Iterator>
    unsigned densify(Iterator begin, Iterator end, std::vector<uint8_t>& bytes, std::vector<char_t>& chars) {
        typedef typename std::make_unsigned<char_t>::type uchar_t;
        chars.clear();
        if (sizeof(char_t) == 1) {
            bool used[256] = {false};
            std::size_t i = 0;
            for (Iterator it = begin; it != end; ++it, ++i) { const uint8_t b = (uint8_t)(uchar_t)*it; bytes[i] = b; used[b] = true; }
            for (int ch = 0; ch < 256; ++ch) if (used[ch]) chars.push_back((char_t)(uchar_t)ch);
            return 1;
        }
        std::vector<uchar_t> u; u.reserve(n);
        for (Iterator it = begin; it != end; ++it) u.push_back((uchar_t)*it);
        std::vector<uchar_t> uniq(u);                   // distinct symbols, ascending by unsigned value
        std::sort(uniq.begin(), uniq.end());
        uniq.erase(std::unique(uniq.begin(), uniq.end()), uniq.end());
        for (std::size_t i = 0; i < uniq.size(); ++i) chars.push_back((char_t)uniq[i]);
        unsigned w = 1;
        while (w < 4 && (uniq.size() - 1) >> (8 * w)) ++w;
        if ((uint64_t)(uniq.size() - 1) >> 32) throw std::runtime_error("psacx: more than 2^32 distinct symbols");
        bytes.resize(n * w);
        for (std::size_t i = 0; i < n; ++i) {
            const uint64_t r = (uint64_t)(std::lower_bound(uniq.begin(), uniq.end(), u[i]) - uniq.begin());
            for (unsigned b = 0; b < w; ++b) bytes[i * w + b] = (uint8_t)(r >> (8 * (w - 1 - b)));
        }
        return w;
    }
    // Results over the byte text of w bytes per symbol -> results over the symbols.  The suffixes that start on a symbol boundary, in
    // the order they have in SA', are SA; the rank among them is ISA; the common prefix of two neighbours among them is the minimum of
    // LCP' over the entries between them, in whole symbols.
    void reduce_wide(unsigned w, const std::vector<index_t>& sa, const std::vector<index_t>& lcp) {
        std::size_t r = 0;
        uint64_t run = ~(uint64_t)0;                    // minimum of LCP' since the last kept entry
        for (std::size_t j = 0; j < sa.size(); ++j) {
            if (!lcp.empty() && j && (uint64_t)lcp[j] < run) run = (uint64_t)lcp[j];
            if ((uint64_t)sa[j] % w) continue;
            const std::size_t i = (std::size_t)((uint64_t)sa[j] / w);
            local_SA[r] = (index_t)i; local_B[i] = (index_t)r;
            if (!lcp.empty()) local_LCP[r] = r ? (index_t)(run / w) : (index_t)0;
            run = ~(uint64_t)0;
            ++r;
        }
    }
    template <typename V> static void dump(const std::string& fn, const std::vector<V>& v) {
        std::ofstream f(fn.c_str(), std::ios::binary | std::ios::trunc);
        f.write(reinterpret_cast<const char*>(v.data()), (std::streamsize)(v.size() * sizeof(V)));
        if (!f) throw std::runtime_error("cannot write " + fn);
    }
    template <typename V> static void slurp(const std::string& fn, std::vector<V>& v) {
        std::ifstream f(fn.c_str(), std::ios::binary | std::ios::ate);
        if (!f) throw std::runtime_error("cannot read " + fn);
        std::size_t bytes = (std::size_t)f.tellg();
        v.resize(bytes / sizeof(V));
        f.seekg(0); f.read(reinterpret_cast<char*>(v.data()), (std::streamsize)(v.size() * sizeof(V)));
    }
};

// construct_suffix_tree(sa, begin, end, comm) of /root/reference/include/suffix_tree.hpp:413-438:
// the (sigma + 1) * n node table (row i = internal node of LCP index i, cell c = child through the
// character with alphabet code c, leaves are n + i, 0 = none).  The text must be bytes.
template <typename Iterator, typename index_t>
std::vector<std::size_t> construct_suffix_tree(suffix_array<char, index_t, true>& sa, Iterator str_begin, Iterator str_end,
                                               const psacx::comm&) {
    static_assert(sizeof(std::size_t) == 8, "size_t must be 64 bit");
    std::vector<uint8_t> text(str_begin, str_end);
    if (text.size() != sa.n) throw std::runtime_error("construct_suffix_tree: text does not match the suffix array");
    uint32_t sigma = 0;
    typedef typename std::conditional<sizeof(index_t) == 4, uint32_t, uint64_t>::type W;
    const W* p_sa = reinterpret_cast<const W*>(sa.local_SA.data());
    const W* p_lcp = reinterpret_cast<const W*>(sa.local_LCP.data());
    struct Call {
        static int run(psacx_ctx* c, const uint8_t* t, uint64_t n, const uint32_t* a, const uint32_t* l, uint64_t* o, uint32_t* s) {
            return psacx_suffix_tree_u32(c, t, n, a, l, o, s);
        }
        static int run(psacx_ctx* c, const uint8_t* t, uint64_t n, const uint64_t* a, const uint64_t* l, uint64_t* o, uint32_t* s) {
            return psacx_suffix_tree_u64(c, t, n, a, l, o, s);
        }
        // p > 1: the table is built by the ranks of the suffix array's communicator (suffix_tree.hpp:440-499)
        static int run(psacx_multi* g, const uint8_t* t, uint64_t n, const uint32_t* a, const uint32_t* l, uint64_t* o, uint32_t* s) {
            return psacx_multi_suffix_tree_u32(g, t, n, a, l, o, s);
        }
        static int run(psacx_multi* g, const uint8_t* t, uint64_t n, const uint64_t* a, const uint64_t* l, uint64_t* o, uint32_t* s) {
            return psacx_multi_suffix_tree_u64(g, t, n, a, l, o, s);
        }
    };
    if (psacx_multi* mg = sa.multi_context()) {
        auto must = [&](int rc) { if (rc != PSACX_OK) throw std::runtime_error(std::string("psacx: ") + psacx_strerror(rc) + " [" + psacx_multi_last_error(mg) + "]"); };
        must(Call::run(mg, text.data(), sa.n, (const W*)nullptr, (const W*)nullptr, nullptr, &sigma));
        std::vector<std::size_t> nodes((std::size_t)(sigma + 1) * sa.n, 0);
        must(Call::run(mg, text.data(), sa.n, p_sa, p_lcp, reinterpret_cast<uint64_t*>(nodes.data()), &sigma));
        return nodes;
    }
    psacx::check(sa.context(), Call::run(sa.context(), text.data(), sa.n, (const W*)nullptr, (const W*)nullptr, nullptr, &sigma));
    std::vector<std::size_t> nodes((std::size_t)(sigma + 1) * sa.n, 0);
    psacx::check(sa.context(), Call::run(sa.context(), text.data(), sa.n, p_sa, p_lcp, reinterpret_cast<uint64_t*>(nodes.data()), &sigma));
    return nodes;
}

#endif // PSACX_SUFFIX_ARRAY_HPP
