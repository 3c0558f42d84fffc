// gsac -- generalized suffix array of the lines of a file, the command line of
// /root/reference/src/gsac.cpp:139-204:  gsac -f <file> [-l] [-c] [-o <basename>] [--device N]
// Strings are the runs between '\n' (src/gsac.cpp:170); positions count the characters with the
// separators left out.  -o (extra) writes <basename>.sa64 / .lcp64 as psac does.
//
// -c: the reference compares with libdivsufsort on the separator-joined text and tolerates
// swapped equal suffixes (src/gsac.cpp:85-135).  Here neighbouring suffixes are compared
// directly, which also pins the order of equal suffixes (text order) and the LCP values.
#include <cstring>
#include <vector>

#include "../../../include/suffix_array.hpp"
#include "bench_common.hpp"

typedef uint64_t index_t;      // src/gsac.cpp:36

template <bool LCP>
static bool check_gsa(const suffix_array<char, index_t, LCP>& sa, const simple_dstringset& ss) {
    const std::size_t n = sa.n;
    std::string cat; cat.reserve(n);
    std::vector<std::size_t> end_of(n);
    for (std::size_t s = 0; s < ss.sizes.size(); ++s) {
        cat.append(ss.str_begins[s], ss.sizes[s]);
        for (std::size_t i = cat.size() - ss.sizes[s]; i < cat.size(); ++i) end_of[i] = cat.size();
    }
    if (sa.local_SA.size() != n) { std::cerr << "[ERROR] GSA has the wrong size" << std::endl; return false; }
    std::vector<bool> seen(n, false);
    for (std::size_t i = 0; i < n; ++i) {
        const std::size_t p = sa.local_SA[i];
        if (p >= n || seen[p]) { std::cerr << "[ERROR] gsa[" << i << "] is not part of a permutation" << std::endl; return false; }
        seen[p] = true;
        if (sa.local_B[p] != i) { std::cerr << "[ERROR] ISA[gsa[" << i << "]] != " << i << std::endl; return false; }
    }
    for (std::size_t i = 1; i < n; ++i) {
        std::size_t a = sa.local_SA[i - 1], b = sa.local_SA[i], c = 0;
        const std::size_t ea = end_of[a], eb = end_of[b];
        while (a + c < ea && b + c < eb && cat[a + c] == cat[b + c]) ++c;
        const bool a_end = a + c == ea, b_end = b + c == eb;
        bool ok;
        if (a_end && b_end) ok = a < b;                      // equal suffixes: text order
        else if (a_end) ok = true;
        else if (b_end) ok = false;
        else ok = (unsigned char)cat[a + c] < (unsigned char)cat[b + c];
        if (!ok) { std::cerr << "[ERROR] gsa[" << i - 1 << "] and gsa[" << i << "] are out of order" << std::endl; return false; }
        if (LCP && sa.local_LCP[i] != c) { std::cerr << "[ERROR] lcp[" << i << "] = " << sa.local_LCP[i] << ", expected " << c << std::endl; return false; }
    }
    if (LCP && n && sa.local_LCP[0] != 0) { std::cerr << "[ERROR] lcp[0] != 0" << std::endl; return false; }
    std::cout << "[SUCCESS] GSA correct" << std::endl;       // src/gsac.cpp:132-134
    return true;
}

template <typename V> static void write_u64(const std::string& fn, const std::vector<V>& v) {
    std::ofstream f(fn.c_str(), std::ios::binary | std::ios::trunc);
    for (std::size_t i = 0; i < v.size(); ++i) { const uint64_t x = (uint64_t)v[i]; f.write(reinterpret_cast<const char*>(&x), 8); }
    if (!f) { std::cerr << "error: cannot write " << fn << std::endl; exit(EXIT_FAILURE); }
}

template <bool LCP>
static int run(const std::string& str, bool check, const std::string& out, int device, const std::vector<int>& devices) {
    simple_dstringset ss(str.begin(), str.end(), psacx::comm(device), '\n');
    if (ss.sum_sizes == 0) { std::cerr << "error: no strings in the input" << std::endl; return EXIT_FAILURE; }
    psacx::alphabet<char> alpha = psacx::alphabet<char>::from_stringset(ss, psacx::comm(device));
    bench_cli::Clock t;
    // --gpus N / --gpus-on-device D,N: the string set is block-distributed over the ranks of the communicator
    suffix_array<char, index_t, LCP> sa(devices.empty() ? psacx::comm(device) : psacx::comm(devices));
    sa.construct_ss(ss, alpha);
    std::cerr << "PSAC time: " << t.elapsed() << " ms" << std::endl;
    if (check && !check_gsa<LCP>(sa, ss)) return 1;
    if (!out.empty()) {
        write_u64(out + ".sa64", sa.local_SA);
        if (LCP) write_u64(out + ".lcp64", sa.local_LCP);
    }
    return 0;
}

int main(int argc, char** argv) {
    bench_cli::Args a(argc, argv, "fo", "lc");
    if (!a.ok || !a.has("-f")) {
        std::cerr << "USAGE: gsac -f <filename> [-l] [-c] [-o <basename>] [--device N] [--gpus N] [--gpus-on-device D,N]\n"
                     "Parallel distributed generalized suffix array and LCP construction (MI355X engine)." << std::endl;
        return EXIT_FAILURE;
    }
    std::string str;
    if (!bench_cli::read_file(a.str("-f"), str)) { std::cerr << "error: cannot open " << a.str("-f") << std::endl; return EXIT_FAILURE; }
    const int device = (int)a.num("--device", 0);
    std::vector<int> devices;
    for (int i = 1; i + 1 < argc; ++i) {
        const std::string f = argv[i], v = argv[i + 1];
        if (f == "--gpus") for (int d = 0; d < atoi(v.c_str()); ++d) devices.push_back(d);
        if (f == "--gpus-on-device") {
            const std::size_t c = v.find(',');
            devices.assign((std::size_t)std::max(c == std::string::npos ? 1 : atoi(v.substr(c + 1).c_str()), 1), atoi(v.substr(0, c).c_str()));
        }
    }
    try {
        return a.has("-l") ? run<true>(str, a.has("-c"), a.str("-o"), device, devices) : run<false>(str, a.has("-c"), a.str("-o"), device, devices);
    } catch (const std::exception& e) {
        std::cerr << "error: " << e.what() << std::endl;
        return EXIT_FAILURE;
    }
}
