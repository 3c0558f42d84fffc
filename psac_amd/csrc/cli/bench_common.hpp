// Shared by the benchmark command lines (benchmark_sac, benchmark_k, benchmark-ansv): argument
// scanning in the shape of the reference's TCLAP definitions, its input generators, a wall clock.
#pragma once
#include <chrono>
#include <cstdlib>
#include <fstream>
#include <iostream>
#include <map>
#include <string>

namespace bench_cli {

// alphabet.hpp:32-45 (glibc rand)
inline std::string rand_dna(std::size_t size, int seed) {
    static const char DNA[4] = {'A', 'C', 'G', 'T'};
    srand(1337 * seed);
    std::string s(size, ' ');
    for (std::size_t i = 0; i < size; ++i) s[i] = DNA[rand() % 4];
    return s;
}

inline bool read_file(const std::string& fn, std::string& out) {
    std::ifstream f(fn.c_str(), std::ios::binary | std::ios::ate);
    if (!f) return false;
    out.resize((std::size_t)f.tellg());
    f.seekg(0); f.read(&out[0], (std::streamsize)out.size());
    return (bool)f;
}

// "-x value" pairs and bare switches; `valued` lists the flags that take a value
struct Args {
    std::map<std::string, std::string> val;
    bool ok;
    Args(int argc, char** argv, const std::string& valued, const std::string& switches) : ok(true) {
        for (int i = 1; i < argc; ++i) {
            std::string a = argv[i];
            if (a.size() == 2 && a[0] == '-' && valued.find(a[1]) != std::string::npos) {
                if (i + 1 >= argc) { std::cerr << "error: missing value for " << a << std::endl; ok = false; return; }
                val[a] = argv[++i];
            } else if (a.size() == 2 && a[0] == '-' && switches.find(a[1]) != std::string::npos) {
                val[a] = "1";
            } else if ((a == "--device" || a == "--gpus" || a == "--gpus-on-device") && i + 1 < argc) {
                val[a] = argv[++i];
            } else { std::cerr << "error: unknown argument " << a << std::endl; ok = false; return; }
        }
    }
    bool has(const std::string& k) const { return val.count(k) != 0; }
    std::string str(const std::string& k, const std::string& d = "") const { return has(k) ? val.find(k)->second : d; }
    long long num(const std::string& k, long long d) const { return has(k) ? atoll(val.find(k)->second.c_str()) : d; }
};

struct Clock {          // mxx::timer: elapsed() in milliseconds
    std::chrono::steady_clock::time_point t0;
    Clock() : t0(std::chrono::steady_clock::now()) {}
    double elapsed() const { return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count(); }
};

} // namespace bench_cli
