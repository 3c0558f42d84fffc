// psac-vs-dss -- the engine against libdivsufsort on the same input, with the flags and the stderr lines of
// /root/reference/src/psac_vs_dss.cpp:59-119:  psac-vs-dss (-f <file> | -r <size>) [-s <seed>] [-c]
//   "PSAC time: <ms> ms", "divsufsort time: <ms> ms"; -c runs sufcheck on both suffix arrays and, beyond the
//   reference, also reports whether the two arrays are identical.
#include "../../include/suffix_array.hpp"
#include "../../psac_amd/csrc/cli/bench_common.hpp"
#include "dss_wrap.hpp"

typedef uint64_t index_t;      // src/psac_vs_dss.cpp:43

int main(int argc, char** argv) {
    bench_cli::Args a(argc, argv, "frs", "c");
    if (!a.ok || a.has("-f") == a.has("-r")) {
        std::cerr << "USAGE: psac-vs-dss {-f <filename>|-r <size>} [-s <int>] [-c] [--device N]\n"
                     "Compare our parallel implementation with divsufsort." << std::endl;
        return EXIT_FAILURE;
    }
    std::string input;
    if (a.has("-f")) {
        if (!bench_cli::read_file(a.str("-f"), input)) { std::cerr << "error: cannot open " << a.str("-f") << std::endl; return EXIT_FAILURE; }
    } else {
        input = bench_cli::rand_dna((std::size_t)a.num("-r", 0), (int)a.num("-s", 0));
    }
    try {
        bench_cli::Clock t;
        double start = t.elapsed();
        suffix_array<char, index_t, false> sa((psacx::comm((int)a.num("--device", 0))));
        sa.construct(input.begin(), input.end(), true);
        std::cerr << "PSAC time: " << t.elapsed() - start << " ms" << std::endl;

        std::vector<index_t> SA;
        start = t.elapsed();
        dss::construct(input, SA);
        std::cerr << "divsufsort time: " << t.elapsed() - start << " ms" << std::endl;

        if (a.has("-c")) {
            std::cerr << "Checking for correctness..." << std::endl;
            if (!dss::check(input, sa.local_SA)) { std::cerr << "ERROR: wrong suffix array from PSAC" << std::endl; return 1; }
            if (!dss::check(input, SA)) std::cerr << "ERROR: wrong suffix array from divsufsort" << std::endl;
            if (sa.local_SA != SA) { std::cerr << "ERROR: PSAC and divsufsort disagree" << std::endl; return 1; }
            std::cerr << "[SUCCESS] PSAC and divsufsort agree" << std::endl;
        }
    } catch (const std::exception& e) {
        std::cerr << "error: " << e.what() << std::endl;
        return EXIT_FAILURE;
    }
    return 0;
}
