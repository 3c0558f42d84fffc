// benchmark-ansv -- all nearest smaller values on generated sequences, as
// /root/reference/src/benchmark_ansv.cpp:232-290: CSV "<n>;<p>;<method>;<milliseconds>".
//   benchmark-ansv -n <size> [-i <iterations>] (-u | -k | -b) [--device N]
// The reference times several exchange strategies of its distributed merge (gansv-allpair,
// -minpair, ...); one rank has no exchange, so the engine reports its single method "gansv-hip".
#include <algorithm>
#include <limits>
#include <vector>

#include "../../../include/psacx.h"
#include "bench_common.hpp"

// src/benchmark_ansv.cpp:171-184 at rank 0 of 1
static std::vector<uint64_t> gen_uniform(std::size_t n) {
    std::vector<uint64_t> v(n);
    std::srand(0);
    std::generate(v.begin(), v.end(), [n]() { return (uint64_t)(std::rand() % n); });
    return v;
}
// src/benchmark_ansv.cpp:186-208: a V-shaped peak around a random minimum
static std::vector<uint64_t> gen_peaks(std::size_t n) {
    std::vector<uint64_t> v(n);
    const std::size_t proc_min = std::rand() % n, n2 = n / 2;
    v[n2] = proc_min;
    for (std::size_t i = 0; i < n2; ++i) v[i] = n - ((n - proc_min) * i / n2);
    for (std::size_t i = n2 + 1; i < n; ++i) v[i] = (n - 2 * proc_min) + ((n - proc_min) * i / n2);
    return v;
}
// src/benchmark_ansv.cpp:210-233 at p = 1: rank 0 belongs to the "second half" (0 >= 0), a
// decreasing sequence of odd values
static std::vector<uint64_t> gen_bitonic(std::size_t n) {
    std::vector<uint64_t> v(n);
    for (std::size_t i = 0; i < n; ++i) v[i] = n - 2 * i + 1;
    return v;
}

int main(int argc, char** argv) {
    bench_cli::Args a(argc, argv, "ni", "kub");
    if (!a.ok || !a.has("-n")) {
        std::cerr << "USAGE: benchmark-ansv -n <size> [-i <num>] {-u|-k|-b} [--device N]" << std::endl;
        return EXIT_FAILURE;
    }
    const std::size_t n = (std::size_t)a.num("-n", 0);
    std::vector<uint64_t> in;
    if (a.has("-k")) in = gen_peaks(n); else if (a.has("-u")) in = gen_uniform(n); else if (a.has("-b")) in = gen_bitonic(n);
    if (in.empty()) return 0;                         // the reference runs on an empty vector here
    psacx_ctx* ctx = nullptr;
    int rc = psacx_create(&ctx, (int)a.num("--device", 0), nullptr);
    if (rc != PSACX_OK) { std::cerr << "error: " << psacx_strerror(rc) << std::endl; return EXIT_FAILURE; }
    std::vector<uint64_t> left(n), right(n);
    for (long long i = 0; i < a.num("-i", 1); ++i) {
        bench_cli::Clock t;
        rc = psacx_ansv_u64(ctx, in.data(), n, 0, 0, std::numeric_limits<uint64_t>::max(), left.data(), right.data());
        if (rc != PSACX_OK) { std::cerr << "error: " << psacx_strerror(rc) << " " << psacx_last_hip_error(ctx) << std::endl; psacx_destroy(ctx); return EXIT_FAILURE; }
        std::cout << n << ";" << 1 << ";" << "gansv-hip" << ";" << t.elapsed() << std::endl;
    }
    psacx_destroy(ctx);
    return 0;
}
