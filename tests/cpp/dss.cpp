// dss -- times libdivsufsort on a file or on the reference's random DNA, with the flags and the output of
// /root/reference/src/dss.cpp:41-84:  dss (-f <file> | -r <size>) [-s <seed>] [-i <iterations>]
// prints "<ms> ms" per iteration on stderr.
#include "../../psac_amd/csrc/cli/bench_common.hpp"
#include "dss_wrap.hpp"

int main(int argc, char** argv) {
    bench_cli::Args a(argc, argv, "frsi", "");
    if (!a.ok || a.has("-f") == a.has("-r")) {
        std::cerr << "USAGE: dss {-f <filename>|-r <size>} [-s <int>] [-i <num>]\n"
                     "Run libdivsufsort suffix array construction and time its execution." << std::endl;
        return EXIT_FAILURE;
    }
    std::string input;
    if (a.has("-f")) {
        if (!bench_cli::read_file(a.str("-f"), input)) { std::cerr << "error: cannot open " << a.str("-f") << std::endl; return EXIT_FAILURE; }
    } else {
        input = bench_cli::rand_dna((std::size_t)a.num("-r", 0), (int)a.num("-s", 0));
    }
    bench_cli::Clock t;
    for (long long i = 0; i < a.num("-i", 1); ++i) {
        std::vector<uint64_t> SA;
        const double start = t.elapsed();
        dss::construct(input, SA);
        std::cerr << t.elapsed() - start << " ms" << std::endl;
    }
    return 0;
}
