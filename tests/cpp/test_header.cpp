// Exercises include/suffix_array.hpp the way the reference's tests use its class
// (test/test_psac.cpp: Mississippi :105, IntAlphabetMiss :277-304, FileIO :306-347).
// Built and run by tests/test_gpu_parity.py::test_cpp_header_program on the GPU box.
#include <cstdio>
#include <cstdlib>
#include <iostream>
#include <string>
#include <vector>

#include "../../include/suffix_array.hpp"

#define CHECK(x) do { if (!(x)) { std::cerr << "FAILED: " #x " at line " << __LINE__ << std::endl; return 1; } } while (0)

int main(int argc, char** argv) {
    const std::string tmp = argc > 1 ? argv[1] : "/tmp";
    const std::vector<uint32_t> exp = {10, 7, 4, 1, 0, 9, 8, 6, 3, 5, 2};
    {
        std::string s = "mississippi";
        suffix_array<char, uint32_t, false> sa((psacx::comm(0)));
        sa.verbose = false;
        sa.construct(s.begin(), s.end());
        CHECK(sa.local_SA == exp);
        CHECK(sa.n == 11 && sa.local_size == 11 && sa.p == 1);
        CHECK(sa.alpha.sigma() == 4 && sa.alpha.bits_per_char() == 3);
        CHECK(sa.local_LCP.empty());
        // repeated construct on the same object (test/test_psac.cpp:148-170)
        sa.construct(s.begin(), s.end(), true, 3);
        CHECK(sa.local_SA == exp);
        sa.construct(s.begin(), s.end(), false, 2);
        CHECK(sa.local_SA == exp);
        // the alphabet and k given by the caller (suffix_array.hpp:365-366); an alphabet that covers more characters than occur
        suffix_array<char, uint32_t, true> sb((psacx::comm(0)));
        sb.verbose = false;
        sb.construct(s.begin(), s.end());
        const std::vector<uint32_t> lcp = sb.local_LCP, isa = sb.local_B;
        const psacx::alphabet<char> wide = psacx::alphabet<char>::from_string(std::string("abcimpswxyz"));
        sb.construct(s.begin(), s.end(), true, wide, 2);
        CHECK(sb.local_SA == exp && sb.local_LCP == lcp && sb.local_B == isa);
        CHECK(sb.alpha == wide && sb.alpha.sigma() == 11);
        sb.construct(s.begin(), s.end(), true, psacx::alphabet<char>::from_string(s), 3);
        CHECK(sb.local_SA == exp && sb.local_LCP == lcp && sb.alpha.sigma() == 4);
        bool threw = false;
        try { sb.construct(s.begin(), s.end(), true, psacx::alphabet<char>::from_string(std::string("ims")), 2); } catch (const std::runtime_error&) { threw = true; }
        CHECK(threw);
    }
    {
        // int alphabet: i = 3, m = 128, p = 66000, s = 12345678
        std::vector<int> v = {128, 3, 12345678, 12345678, 3, 12345678, 12345678, 3, 66000, 66000, 3};
        suffix_array<int, unsigned int, true> sa((psacx::comm(0)));
        sa.verbose = false;
        sa.construct(v.begin(), v.end());
        CHECK(sa.local_SA == exp);
        const std::vector<unsigned int> lcp = {0, 1, 1, 4, 0, 0, 1, 0, 2, 1, 3};
        CHECK(sa.local_LCP == lcp);
    }
    {
        // write / read round trip (suffix_array.hpp:232-265): raw little-endian arrays
        std::string s;
        srand(7);
        for (int i = 0; i < 50000; ++i) s.push_back("ACGT"[rand() % 4]);
        suffix_array<char, uint64_t, true> sa((psacx::comm(0)));
        sa.verbose = false;
        sa.construct(s.begin(), s.end());
        sa.write(tmp + "/psacx_fileio");
        suffix_array<char, uint64_t, true> sb((psacx::comm(0)));
        sb.read(tmp + "/psacx_fileio");
        CHECK(sb.local_SA == sa.local_SA);
        CHECK(sb.local_LCP == sa.local_LCP);
        CHECK(sb.n == 50000);
        FILE* f = fopen((tmp + "/psacx_fileio.alpha").c_str(), "rb");
        CHECK(f != nullptr);
        char buf[8]; size_t got = fread(buf, 1, 8, f); fclose(f);
        CHECK(got == 4 && std::string(buf, 4) == "ACGT");
    }
    {
        // FileIO with left-branching characters (test/test_psac.cpp:306-347: suffix_array<char, size_t, true, true>)
        std::string s = "mississippi";
        suffix_array<char, size_t, true, true> sa((psacx::comm(0)));
        sa.verbose = false;
        sa.construct(s.begin(), s.end());
        const std::string lc("\0\0ppimippip", 11);
        CHECK(std::string(sa.local_Lc.begin(), sa.local_Lc.end()) == lc);
        sa.write(tmp + "/miss");
        suffix_array<char, size_t, true, true> sa2((psacx::comm(0)));
        sa2.read(tmp + "/miss");
        CHECK(sa.local_SA == sa2.local_SA);
        CHECK(sa.local_LCP == sa2.local_LCP);
        CHECK(sa.local_Lc == sa2.local_Lc);
        CHECK(sa.alpha == sa2.alpha);
        // int symbols map back to the caller's values
        std::vector<int> v = {128, 3, 12345678, 12345678, 3, 12345678, 12345678, 3, 66000, 66000, 3};
        suffix_array<int, uint32_t, true, true> si((psacx::comm(0)));
        si.verbose = false;
        si.construct(v.begin(), v.end());
        const std::vector<int> ilc = {0, 0, 66000, 66000, 3, 128, 3, 66000, 66000, 3, 66000};
        CHECK(si.local_Lc == ilc);
    }
    {
        // more than 256 distinct symbols (the reference's alphabet<int> is unbounded, alphabet.hpp:205-236): the text goes to the
        // engine as two (three) bytes per symbol.  Expected arrays: comparison sort of the suffixes, common prefixes counted directly.
        for (int variant = 0; variant < 3; ++variant) {
            const std::size_t m = variant == 2 ? 3000 : 20000;
            const unsigned distinct = variant == 0 ? 1000u : variant == 1 ? 300u : 70000u;
            std::vector<int> v(m);
            uint64_t x = 88172645463325252ull + variant;
            for (std::size_t i = 0; i < m; ++i) {
                x ^= x << 13; x ^= x >> 7; x ^= x << 17;
                // (repeats, so that common prefixes of several symbols occur; negative values sort above the positive ones: unsigned order)
                v[i] = (i >= 500 && (x >> 60) < 3) ? v[i - 500] : (int)((x >> 20) % distinct) * (variant == 1 ? -7919 : 7919) - 5;
            }
            if (variant == 2) for (std::size_t i = 0; i < m; ++i) v[i] = (int)(i * 23u % distinct);          // all 70000 > 2^16 values cannot occur in 3000: force them
            if (variant == 2) { v.resize(70000 + 3000); for (std::size_t i = 3000; i < v.size(); ++i) v[i] = (int)(i - 3000); }
            const std::size_t nn = v.size();
            std::vector<uint32_t> esa(nn), eisa(nn), elcp(nn, 0);
            for (std::size_t i = 0; i < nn; ++i) esa[i] = (uint32_t)i;
            std::sort(esa.begin(), esa.end(), [&](uint32_t a, uint32_t b) {
                return std::lexicographical_compare(v.begin() + a, v.end(), v.begin() + b, v.end(),
                                                    [](int c, int d) { return (unsigned)c < (unsigned)d; });
            });
            for (std::size_t r = 0; r < nn; ++r) eisa[esa[r]] = (uint32_t)r;
            for (std::size_t r = 1; r < nn; ++r) {
                std::size_t a = esa[r - 1], b = esa[r], l = 0;
                while (a + l < nn && b + l < nn && v[a + l] == v[b + l]) ++l;
                elcp[r] = (uint32_t)l;
            }
            suffix_array<int, uint32_t, true, true> sw((psacx::comm(0)));
            sw.verbose = false;
            sw.construct(v.begin(), v.end());
            CHECK(sw.n == nn && sw.local_SA.size() == nn);
            CHECK(sw.alpha.sigma() > 256);
            CHECK(sw.local_SA == esa);
            CHECK(sw.local_B == eisa);
            CHECK(sw.local_LCP == elcp);
            for (std::size_t r = 1; r < nn; ++r) {
                const std::size_t at = (std::size_t)esa[r - 1] + elcp[r];
                CHECK(sw.local_Lc[r] == (at < nn ? v[at] : 0));
            }
            suffix_array<int, uint64_t, false> s64((psacx::comm(0)));
            s64.verbose = false;
            s64.construct(v.begin(), v.end(), true, 2);
            CHECK(s64.local_LCP.empty());
            for (std::size_t r = 0; r < nn; ++r) CHECK(s64.local_SA[r] == esa[r] && s64.local_B[r] == eisa[r]);
        }
    }
    {
        // generalized suffix array (test/test_gsa.cpp:73-105, SimpleTiny)
        std::vector<std::string> strs = {"abab", "baba"};
        std::string flat = flatten_strings(strs);
        simple_dstringset ss(flat.begin(), flat.end(), psacx::comm(0));
        CHECK(ss.sizes.size() == 2 && ss.sum_sizes == 8);
        psacx::alphabet<char> a = psacx::alphabet<char>::from_string("ab", psacx::comm(0));
        suffix_array<char, uint64_t, true> sa((psacx::comm(0)));
        sa.verbose = false;
        sa.construct_ss(ss, a);
        const std::vector<uint64_t> ex_gsa = {7, 2, 5, 0, 3, 6, 1, 4}, ex_lcp = {0, 1, 2, 3, 0, 1, 2, 3};
        CHECK(sa.local_SA == ex_gsa);
        CHECK(sa.local_LCP == ex_lcp);
        CHECK(sa.alpha.sigma() == 2);
    }
    {
        // suffix tree node table (test/test_suffixtree.cpp:68-83)
        std::string s = "mississippi";
        suffix_array<char, uint64_t, true> sa((psacx::comm(0)));
        sa.verbose = false;
        sa.construct(s.begin(), s.end());
        std::vector<std::size_t> nodes = construct_suffix_tree(sa, s.begin(), s.end(), psacx::comm(0));
        const std::vector<std::size_t> solution = {0, 1, 15, 6, 9, 11, 0, 0, 12, 3, 0, 0, 0, 0, 0, 0, 0, 0, 13, 14, 0, 0, 0, 0, 0,
                                                   0, 0, 0, 0, 0, 0, 16, 0, 17, 0, 0, 0, 0, 0, 0, 0, 0, 0, 18, 19, 0, 8, 0, 0, 10,
                                                   0, 0, 0, 20, 21};
        CHECK(nodes == solution);
    }
    {
        // errors surface as std::runtime_error (suffix_array.hpp:226-227)
        std::string empty;
        suffix_array<char, uint32_t, true> sa((psacx::comm(0)));
        bool threw = false;
        try { sa.construct(empty.begin(), empty.end()); } catch (const std::runtime_error&) { threw = true; }
        CHECK(threw);
    }
    std::cout << "cpp header tests passed" << std::endl;
    return 0;
}
