// check.hpp -- the `-c` verification of the psac CLI (host side, verification only; the
// construction itself never runs on the CPU).  Follows /root/reference/include/
// check_suffix_array.hpp:56-88 (check_SA), :106-126 (check_lcp against lcp.hpp:46-77).
#pragma once
#include <cstdint>
#include <iostream>
#include <string>
#include <vector>

namespace psacx_cli {

template <typename index_t>
bool check_SA(const std::string& s, const std::vector<index_t>& SA, const std::vector<index_t>& ISA) {
    const std::size_t n = s.size();
    if (SA.size() != n || ISA.size() != n) { std::cerr << "[ERROR] size mismatch" << std::endl; return false; }
    for (std::size_t i = 0; i < n; ++i) {
        if ((std::size_t)SA[i] >= n) { std::cerr << "[ERROR] SA[" << i << "] out of range" << std::endl; return false; }
        if ((std::size_t)ISA[SA[i]] != i) { std::cerr << "[ERROR] ISA[SA[" << i << "]] != " << i << std::endl; return false; }
    }
    for (std::size_t i = 1; i < n; ++i) {
        const std::size_t a = SA[i - 1], b = SA[i];
        const unsigned char ca = (unsigned char)s[a], cb = (unsigned char)s[b];
        bool ok = ca < cb;
        if (ca == cb) ok = (a + 1 == n) || (b + 1 < n && ISA[a + 1] < ISA[b + 1]);
        if (!ok) { std::cerr << "[ERROR] wrong suffix order at SA position " << i << std::endl; return false; }
    }
    return true;
}

template <typename index_t>
bool check_lcp(const std::string& s, const std::vector<index_t>& SA, const std::vector<index_t>& ISA,
               const std::vector<index_t>& LCP) {
    const std::size_t n = s.size();
    if (LCP.size() != n) { std::cerr << "[ERROR] LCP size mismatch" << std::endl; return false; }
    if (n && LCP[0] != 0) { std::cerr << "[ERROR] LCP[0] != 0" << std::endl; return false; }
    std::size_t h = 0;
    for (std::size_t i = 0; i < n; ++i) {
        const std::size_t r = ISA[i];
        if (r == 0) { h = 0; continue; }
        const std::size_t j = SA[r - 1];
        if (h > 0) --h;
        while (i + h < n && j + h < n && s[i + h] == s[j + h]) ++h;
        if ((std::size_t)LCP[r] != h) {
            std::cerr << "[ERROR] LCP[" << r << "] = " << LCP[r] << ", expected " << h << std::endl;
            return false;
        }
    }
    return true;
}

} // namespace psacx_cli
