// psac -- command line of the MI355X engine with the reference's flags and output files
// (/root/reference/src/psac.cpp:56-153):
//   psac (-f <file> | -r <size>) [-s <seed>] [-o <basename>] [-l] [-t] [-c]
// writes <basename>.sa64 (and .lcp64 with -l) as raw little-endian uint64 arrays and prints
// "PSAC time: <ms> ms".  Extra flags: --device N, --index {32,64,auto} (files stay uint64), --gpus N: the text is
// block-decomposed over GPUs 0..N-1 (src/psac.cpp:85-93 over MPI ranks) and built by the multi-GPU engine;
// --gpus-on-device D,N: N ranks sharing device D (how a one-GPU box exercises that path).
#include <chrono>
#include <cstdlib>
#include <cstring>
#include <fstream>
#include <iostream>
#include <string>
#include <vector>

#include "../../../include/suffix_array.hpp"
#include "check.hpp"

typedef uint64_t index_t;     // src/psac.cpp:54

static std::string rand_dna(std::size_t size, int seed) {   // alphabet.hpp:32-45
    static const char DNA[4] = {'A', 'C', 'G', 'T'};
    srand(1337 * seed);
    std::string s(size, ' ');
    for (std::size_t i = 0; i < size; ++i) s[i] = DNA[rand() % 4];
    return s;
}

static void usage() {
    std::cerr << "USAGE: psac {-f <filename>|-r <size>} [-s <int>] [-o <filename>] [-l] [-t] [-c] [--device N] [--gpus N] [--index 32|64|auto]\n"
                 "Parallel distributed suffix array and LCP construction (MI355X engine).\n";
}

template <typename T> static void write_u64(const std::string& fn, const std::vector<T>& v) {   // mxx::write_ordered, src/psac.cpp:127-128
    std::ofstream f(fn.c_str(), std::ios::binary | std::ios::trunc);
    std::vector<uint64_t> buf(1 << 16);
    for (std::size_t i = 0; i < v.size(); i += buf.size()) {
        const std::size_t m = std::min(buf.size(), v.size() - i);
        for (std::size_t j = 0; j < m; ++j) buf[j] = (uint64_t)v[i + j];
        f.write(reinterpret_cast<const char*>(buf.data()), (std::streamsize)(m * 8));
    }
    if (!f) { std::cerr << "error: cannot write " << fn << std::endl; exit(EXIT_FAILURE); }
}

// src/psac.cpp:96-114: SA + LCP, then the suffix-tree node table (ANSV over LCP inside)
template <typename IT>
static void tree_step(suffix_array<char, IT, true>& sa, const std::string& str, const std::string& out, int device, double ms) {
    auto t1 = std::chrono::steady_clock::now();
    std::vector<std::size_t> nodes = construct_suffix_tree(sa, str.begin(), str.end(), psacx::comm(device));
    double ms2 = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t1).count();
    std::size_t edges = 0;
    for (std::size_t i = 0; i < nodes.size(); ++i) edges += nodes[i] != 0;
    std::cerr << "ST time: " << ms2 << " ms" << std::endl;
    std::cerr << "Total  : " << ms + ms2 << " ms" << std::endl;
    std::cerr << "ST edges: " << edges << std::endl;
    if (!out.empty()) std::cerr << "Error, output of ST not supported" << std::endl;
}
template <typename IT>
static void tree_step(suffix_array<char, IT, false>&, const std::string&, const std::string&, int, double) {}

static std::vector<int> g_devices;      // --gpus: the devices of the communicator (empty: the single --device)

template <typename IT, bool LCP>
static int run(const std::string& str, const std::string& out, bool check, int device, bool tree) {
    suffix_array<char, IT, LCP> sa(g_devices.empty() ? psacx::comm(device) : psacx::comm(g_devices));
    auto t0 = std::chrono::steady_clock::now();
    sa.construct(str.begin(), str.end(), true);
    double ms = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
    std::cerr << (tree ? "SA time: " : "PSAC time: ") << ms << " ms" << std::endl;
    if (tree) tree_step(sa, str, out, device, ms);
    if (check) {
        bool ok = psacx_cli::check_SA(str, sa.local_SA, sa.local_B);
        if (ok && LCP) ok = psacx_cli::check_lcp(str, sa.local_SA, sa.local_B, sa.local_LCP);
        if (!ok) { std::cerr << "[ERROR] Test unsuccessful" << std::endl; return 1; }
        std::cerr << "[SUCCESS] Suffix Array" << (LCP ? " and LCP" : "") << " are correct" << std::endl;
    }
    if (!out.empty() && !tree) {
        write_u64(out + ".sa64", sa.local_SA);
        if (LCP) write_u64(out + ".lcp64", sa.local_LCP);
    }
    return 0;
}

int main(int argc, char** argv) {
    std::string file, out, index = "auto";
    std::size_t rsize = 0; bool have_r = false, lcp = false, tree = false, check = false;
    int seed = 0, device = 0;
    for (int i = 1; i < argc; ++i) {
        const std::string a = argv[i];
        auto need = [&](const char* name) -> const char* {
            if (i + 1 >= argc) { std::cerr << "error: missing value for " << name << std::endl; usage(); exit(EXIT_FAILURE); }
            return argv[++i];
        };
        if (a == "-f" || a == "--file") file = need("-f");
        else if (a == "-r" || a == "--random") { rsize = (std::size_t)strtoull(need("-r"), nullptr, 10); have_r = true; }
        else if (a == "-s" || a == "--seed") seed = atoi(need("-s"));
        else if (a == "-o" || a == "--outfile") out = need("-o");
        else if (a == "-l" || a == "--lcp") lcp = true;
        else if (a == "-t" || a == "--tree") tree = true;
        else if (a == "-c" || a == "--check") check = true;
        else if (a == "--device") device = atoi(need("--device"));
        else if (a == "--gpus") { const int g = atoi(need("--gpus")); g_devices.clear(); for (int d = 0; d < g; ++d) g_devices.push_back(d); }
        else if (a == "--gpus-on-device") {
            const std::string v = need("--gpus-on-device");
            const std::size_t c = v.find(',');
            const int d = atoi(v.substr(0, c).c_str()), g = c == std::string::npos ? 1 : atoi(v.substr(c + 1).c_str());
            g_devices.assign((std::size_t)std::max(g, 1), d);
        }
        else if (a == "--index") index = need("--index");
        else if (a == "-h" || a == "--help") { usage(); return 0; }
        else { std::cerr << "error: unknown argument " << a << std::endl; usage(); return EXIT_FAILURE; }
    }
    if (file.empty() == !have_r) {    // TCLAP xorAdd, src/psac.cpp:67-69
        std::cerr << "error: exactly one of -f and -r is required" << std::endl; usage(); return EXIT_FAILURE;
    }
    std::string str;
    if (!file.empty()) {
        std::ifstream f(file.c_str(), std::ios::binary | std::ios::ate);
        if (!f) { std::cerr << "error: cannot open " << file << std::endl; return EXIT_FAILURE; }
        str.resize((std::size_t)f.tellg());
        f.seekg(0); f.read(&str[0], (std::streamsize)str.size());
    } else {
        str = rand_dna(rsize, seed);
    }
    const bool use32 = index == "32" || (index == "auto" && str.size() < 0xFFFFFFFEull);
    try {
        if (tree) return use32 ? run<uint32_t, true>(str, out, check, device, true) : run<uint64_t, true>(str, out, check, device, true);
        if (lcp) return use32 ? run<uint32_t, true>(str, out, check, device, false) : run<uint64_t, true>(str, out, check, device, false);
        return use32 ? run<uint32_t, false>(str, out, check, device, false) : run<uint64_t, false>(str, out, check, device, false);
    } catch (const std::exception& e) {
        std::cerr << "error: " << e.what() << std::endl;
        return EXIT_FAILURE;
    }
}
