// print64 -- dumps a raw little-endian uint64 file as decimal lines
// (/root/reference/src/print64.cpp, README.md:84-100).
#include <cstdint>
#include <fstream>
#include <iostream>
int main(int argc, char** argv) {
    if (argc < 2) { std::cerr << "Usage: " << argv[0] << " <file>" << std::endl; return 1; }
    std::ifstream f(argv[1], std::ios::binary);
    if (!f) { std::cerr << "cannot open " << argv[1] << std::endl; return 1; }
    uint64_t x;
    while (f.read(reinterpret_cast<char*>(&x), sizeof(x))) std::cout << x << "\n";
    return 0;
}
