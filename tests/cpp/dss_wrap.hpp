// dss_wrap.hpp -- C++ calls into libdivsufsort (the oracle/_ref build of /root/reference/ext/libdivsufsort) for
// the two comparison command lines, after the interface of /root/reference/include/divsufsort_wrapper.hpp:54-100
// (dss::construct picks divsufsort / divsufsort64 by index width; dss::check wraps sufcheck).
// Test infrastructure: these tools link the CPU checker and therefore live under tests/, not in the product.
#pragma once
#include <divsufsort.h>
#include <divsufsort64.h>

#include <limits>
#include <stdexcept>
#include <string>
#include <vector>

namespace dss {

template <typename T>
void construct(const std::string& s, std::vector<T>& SA) {
    const std::size_t n = s.size();
    SA.resize(n);
    if (n == 0) return;
    const sauchar_t* t = reinterpret_cast<const sauchar_t*>(s.data());
    if (sizeof(T) == sizeof(saidx_t)) {
        if (n >= (std::size_t)std::numeric_limits<saidx_t>::max()) throw std::runtime_error("Input size is too large for 32bit indexing.");
        if (divsufsort(t, reinterpret_cast<saidx_t*>(&SA[0]), (saidx_t)n) != 0) throw std::runtime_error("divsufsort failed");
    } else {
        if (divsufsort64(t, reinterpret_cast<saidx64_t*>(&SA[0]), (saidx64_t)n) != 0) throw std::runtime_error("divsufsort64 failed");
    }
}

template <typename T>
bool check(const std::string& s, const std::vector<T>& SA) {
    const sauchar_t* t = reinterpret_cast<const sauchar_t*>(s.data());
    if (sizeof(T) == sizeof(saidx_t)) return sufcheck(t, reinterpret_cast<const saidx_t*>(&SA[0]), (saidx_t)s.size(), 0) == 0;
    return sufcheck64(t, reinterpret_cast<const saidx64_t*>(&SA[0]), (saidx64_t)s.size(), 0) == 0;
}

} // namespace dss
