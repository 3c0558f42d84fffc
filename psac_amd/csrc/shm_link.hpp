// shm_link.hpp -- host-staged transport between the ranks of one node that live in different processes.
//
// psac runs one MPI rank per block (src/psac.cpp:85-93); without a GPU-aware MPI its exchanges stage through host
// memory.  This link is that deployment for the multi-GPU engine: every rank is its own process (psacx_multi_create_rank
// with PSACX_MULTI_TRANSPORT=shm), all on one host, and the exchanges of multi.hpp go device -> a POSIX shared-memory
// segment -> device.  It exists so that the process-per-GPU bookkeeping (L = 1 < P: count all-gathers, rank(i) != i,
// receive offsets by source) runs on a box with ONE GPU, where RCCL refuses two ranks on the same device; over xGMI the
// same code paths run with RCCL.  The segment is named after the communicator id the host broadcasts.
//
// Layout: header | P scalar slots | P data boxes.  All synchronisation is a sense-reversing barrier on two words of
// the header (lock-free atomics work across processes on memory both have mapped).
#pragma once
#include <fcntl.h>
#include <sched.h>
#include <sys/mman.h>
#include <sys/stat.h>
#include <unistd.h>

#include <atomic>
#include <chrono>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>

namespace psacx {

struct ShmLink {
    struct Header {
        std::atomic<uint32_t> magic, arrived, sense, attached;
        uint32_t nranks, pad;
        uint64_t slot_bytes, box_bytes;
    };
    static constexpr uint32_t MAGIC = 0x70736158u;      // "psaX"
    std::string name;
    char* base = nullptr;
    size_t bytes = 0;
    int rank = 0, P = 1;
    uint32_t my_sense = 0;
    size_t slot_bytes = 0, box_bytes = 0;
    double timeout_s = 120.0;

    Header* hdr() const { return reinterpret_cast<Header*>(base); }
    char* slot(int r) const { return base + 4096 + (size_t)r * slot_bytes; }
    char* box(int r) const { return base + 4096 + (size_t)P * slot_bytes + (size_t)r * box_bytes; }

    static std::string name_of(const void* id128) {
        // the first bytes of an RCCL unique id are an address that repeats between communicators: hash all 128
        const unsigned char* p = static_cast<const unsigned char*>(id128);
        uint64_t h = 1469598103934665603ull;
        for (int i = 0; i < 128; ++i) { h ^= p[i]; h *= 1099511628211ull; }
        char buf[64];
        snprintf(buf, sizeof(buf), "/psacx_%016llx", (unsigned long long)h);
        return buf;
    }

    // id128 must be unique per communicator (an RCCL unique id is).  A segment of the same name that a crashed run left behind, or that
    // rank 0 has not yet replaced, is recognised and dropped: a rank that is not rank 0 only stays on a segment whose name still leads to
    // the inode it mapped, waits there until all P ranks have attached, and refuses one that P ranks hold already.
    bool open(int rank_, int nranks, const void* id128, std::string& err, size_t box = 0) {
        rank = rank_; P = nranks; name = name_of(id128);
        opened = false;
        slot_bytes = (size_t)1 << 20;
        box_bytes = box ? box : (size_t)32 << 20;          // (psacx_multi_create_rank_ex: shm_box_bytes)
        box_bytes = (box_bytes + 4095) & ~(size_t)4095;
        if (box_bytes < 4096) box_bytes = 4096;
        bytes = 4096 + (size_t)P * (slot_bytes + box_bytes);
        const auto t0 = std::chrono::steady_clock::now();
        auto late = [&]() { return std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count() > timeout_s; };
        if (rank == 0) {
            (void)shm_unlink(name.c_str());
            const int fd = shm_open(name.c_str(), O_CREAT | O_EXCL | O_RDWR, 0600);
            if (fd < 0 || ftruncate(fd, (off_t)bytes) != 0) { err = "shm_open / ftruncate of " + name + " failed"; if (fd >= 0) close(fd); return false; }
            base = static_cast<char*>(mmap(nullptr, bytes, PROT_READ | PROT_WRITE, MAP_SHARED, fd, 0));
            close(fd);
            if (base == MAP_FAILED) { base = nullptr; err = "mmap of " + name + " failed"; return false; }
            Header* h = hdr();
            h->arrived.store(0); h->sense.store(0); h->attached.store(1);
            h->nranks = (uint32_t)P; h->slot_bytes = slot_bytes; h->box_bytes = box_bytes;
            h->magic.store(MAGIC, std::memory_order_release);
            while (h->attached.load(std::memory_order_acquire) < (uint32_t)P) {
                if (late()) { err = "timed out waiting for the peers of " + name; munmap(base, bytes); base = nullptr; (void)shm_unlink(name.c_str()); return false; }
                usleep(200);
            }
        } else {
            for (;;) {
                int fd = -1;
                struct stat st;
                for (;;) {
                    fd = shm_open(name.c_str(), O_RDWR, 0600);
                    if (fd >= 0 && fstat(fd, &st) == 0 && (size_t)st.st_size >= bytes) break;
                    if (fd >= 0) { close(fd); fd = -1; }
                    if (late()) { err = "timed out waiting for " + name; return false; }
                    usleep(1000);
                }
                const ino_t ino = st.st_ino;
                base = static_cast<char*>(mmap(nullptr, bytes, PROT_READ | PROT_WRITE, MAP_SHARED, fd, 0));
                close(fd);
                if (base == MAP_FAILED) { base = nullptr; err = "mmap of " + name + " failed"; return false; }
                Header* h = hdr();
                // the name must still lead to this inode while we wait on it (rank 0 unlinks whatever was there and makes its own)
                auto stale = [&]() {
                    const int f2 = shm_open(name.c_str(), O_RDWR, 0600);
                    struct stat s2;
                    const bool gone = f2 < 0 || fstat(f2, &s2) != 0 || s2.st_ino != ino;
                    if (f2 >= 0) close(f2);
                    return gone;
                };
                // (a segment that looks wrong -- another shape, P ranks on it already -- may be one a crashed run left behind and rank 0 is
                //  about to replace: it is only an error if the name still leads to it when the time is up)
                bool retry = false, joined = false;
                const char* wrong = nullptr;
                for (unsigned spin = 0;; ++spin) {
                    if (!joined && !wrong && h->magic.load(std::memory_order_acquire) == MAGIC) {
                        if (h->nranks != (uint32_t)P || h->box_bytes != box_bytes) wrong = " was made for another communicator shape";
                        else if (h->attached.fetch_add(1, std::memory_order_acq_rel) >= (uint32_t)P) wrong = " is held by all its ranks already: the communicator id must be unique";
                        else joined = true;
                    }
                    if (joined && h->attached.load(std::memory_order_acquire) >= (uint32_t)P) {
                        // (a segment a crashed run left behind can reach its count through late joiners like this one while rank 0 is
                        //  replacing it: the name must still lead here before the count is believed -- ADVICE r4)
                        if (stale()) { retry = true; }
                        break;
                    }
                    if ((wrong || (spin & 63u) == 63u) && stale()) { retry = true; break; }
                    if (late()) {
                        err = wrong ? "shared segment " + name + wrong : "timed out waiting for the peers of " + name;
                        munmap(base, bytes); base = nullptr; return false;
                    }
                    usleep(wrong ? 2000 : 200);
                }
                if (!retry) break;
                munmap(base, bytes); base = nullptr;           // a segment somebody left behind: look again
                usleep(1000);
            }
        }
        if (!barrier(err)) { munmap(base, bytes); base = nullptr; if (rank == 0) (void)shm_unlink(name.c_str()); return false; }
        opened = true;
        return true;
    }
    bool opened = false;           // open() went through: the closing barrier has peers

    // false on timeout (a peer died): the caller turns it into an error instead of hanging
    bool barrier(std::string& err) {
        Header* h = hdr();
        my_sense ^= 1u;
        if (h->arrived.fetch_add(1, std::memory_order_acq_rel) + 1 == (uint32_t)P) {
            h->arrived.store(0, std::memory_order_relaxed);
            h->sense.store(my_sense, std::memory_order_release);
            return true;
        }
        const auto t0 = std::chrono::steady_clock::now();
        unsigned spins = 0;
        while (h->sense.load(std::memory_order_acquire) != my_sense) {
            if (++spins > 64) { sched_yield(); }
            if ((spins & 0xFFFu) == 0 && std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count() > timeout_s) {
                err = "shared-memory barrier timed out (a peer rank is gone?)";
                return false;
            }
        }
        return true;
    }

    void close_link() {
        if (!base) return;
        std::string err;
        if (opened) (void)barrier(err);                   // nobody unmaps while a peer still reads (a link whose open() failed has no peers to wait for)
        munmap(base, bytes);
        base = nullptr;
        if (rank == 0 && opened) (void)shm_unlink(name.c_str());
        opened = false;
    }
};

} // namespace psacx
