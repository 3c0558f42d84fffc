// benchmark_k -- construction time for a given initial k-mer length, as
// /root/reference/src/benchmark_k.cpp:35-67: CSV "<p>;<method>;<k>;<milliseconds>".
//   benchmark_k (-f <file> | -r <size>) [-i <iterations>] [-k <k>] [--device N]
#include <vector>

#include "../../../include/suffix_array.hpp"
#include "bench_common.hpp"

static void one(const std::string& str, bool fast, int k, const char* method, int device) {
    bench_cli::Clock t;
    suffix_array<char, std::size_t, false> sa((psacx::comm(device)));
    sa.verbose = false;
    sa.construct(str.begin(), str.end(), fast, (unsigned int)k);
    std::cout << 1 << ";" << method << ";" << k << ";" << t.elapsed() << std::endl;
}

int main(int argc, char** argv) {
    bench_cli::Args a(argc, argv, "frik", "");
    if (!a.ok || a.has("-f") == a.has("-r")) {
        std::cerr << "USAGE: benchmark_k {-f <filename>|-r <size>} [-i <num>] [-k <size>] [--device N]" << std::endl;
        return EXIT_FAILURE;
    }
    std::string str;
    if (a.has("-f")) { if (!bench_cli::read_file(a.str("-f"), str)) { std::cerr << "error: cannot open " << a.str("-f") << std::endl; return EXIT_FAILURE; } }
    else str = bench_cli::rand_dna((std::size_t)a.num("-r", 0), 0);
    const int device = (int)a.num("--device", 0), k = (int)a.num("-k", 0);
    try {
        for (long long i = 0; i < a.num("-i", 1); ++i) {
            one(str, true, k, "reg-fast-nolcp", device);
            one(str, false, k, "reg-nolcp", device);
        }
    } catch (const std::exception& e) { std::cerr << "error: " << e.what() << std::endl; return EXIT_FAILURE; }
    return 0;
}
