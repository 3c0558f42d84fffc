// benchmark_sac -- the four construction variants of /root/reference/src/benchmark.cpp:35-80 on
// the MI355X engine, one CSV line each: "<p>;<method>;<milliseconds>".
//   benchmark_sac (-f <file> | -r <size>) [-i <iterations>] [--device N]
#include <vector>

#include "../../../include/suffix_array.hpp"
#include "bench_common.hpp"

template <bool LCP>
static void one(const std::string& str, bool fast, const char* method, int device) {
    bench_cli::Clock t;
    suffix_array<char, std::size_t, LCP> sa((psacx::comm(device)));
    sa.verbose = false;
    sa.construct(str.begin(), str.end(), fast);
    std::cout << 1 << ";" << method << ";" << t.elapsed() << std::endl;
}

int main(int argc, char** argv) {
    bench_cli::Args a(argc, argv, "fri", "");
    if (!a.ok || a.has("-f") == a.has("-r")) {
        std::cerr << "USAGE: benchmark_sac {-f <filename>|-r <size>} [-i <num>] [--device N]" << std::endl;
        return EXIT_FAILURE;
    }
    std::string str;
    if (a.has("-f")) { if (!bench_cli::read_file(a.str("-f"), str)) { std::cerr << "error: cannot open " << a.str("-f") << std::endl; return EXIT_FAILURE; } }
    else str = bench_cli::rand_dna((std::size_t)a.num("-r", 0), 0);       // src/benchmark.cpp:137 (seed = rank)
    const int device = (int)a.num("--device", 0);
    try {
        for (long long i = 0; i < a.num("-i", 1); ++i) {
            one<false>(str, false, "reg-nolcp", device);
            one<false>(str, true, "reg-fast-nolcp", device);
            one<true>(str, false, "reg-lcp", device);
            one<true>(str, true, "reg-fast-lcp", device);
        }
    } catch (const std::exception& e) { std::cerr << "error: " << e.what() << std::endl; return EXIT_FAILURE; }
    return 0;
}
