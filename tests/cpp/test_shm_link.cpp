// The host-staged transport of the process-per-rank deployment (psac_amd/csrc/shm_link.hpp) without a GPU: P forked
// processes attach to one segment, all-gather words through the slots and stream a buffer through the boxes in rounds,
// the way MultiRun::gather / exchange use the link.  Exit code 0 = every rank saw what it should.
#include <sys/wait.h>
#include <vector>
#include "../../psac_amd/csrc/shm_link.hpp"

using psacx::ShmLink;

static int run_rank(int rank, int P, const unsigned char* id) {
    ShmLink sh;
    std::string err;
    sh.timeout_s = 30;
    if (!sh.open(rank, P, id, err, 65536)) { fprintf(stderr, "rank %d: %s\n", rank, err.c_str()); return 2; }
    int bad = 0;
    for (int round = 0; round < 50; ++round) {
        // scalar all-gather
        uint64_t mine[3] = {(uint64_t)rank * 1000 + round, (uint64_t)round, 7};
        std::memcpy(sh.slot(rank), mine, sizeof(mine));
        if (!sh.barrier(err)) return 3;
        for (int r = 0; r < P; ++r) {
            uint64_t got[3];
            std::memcpy(got, sh.slot(r), sizeof(got));
            if (got[0] != (uint64_t)r * 1000 + round || got[1] != (uint64_t)round || got[2] != 7) ++bad;
        }
        if (!sh.barrier(err)) return 3;
    }
    // a stream longer than a box, in rounds: rank r sends element i = r * 2^20 + i, everybody reads everybody's
    const size_t len = sh.box_bytes / 4 * 2 + 123;
    std::vector<uint32_t> src(len);
    for (size_t i = 0; i < len; ++i) src[i] = (uint32_t)(rank << 20) + (uint32_t)i;
    const size_t per = sh.box_bytes / 4;
    for (size_t w0 = 0; w0 < len; w0 += per) {
        const size_t cnt = std::min(per, len - w0);
        std::memcpy(sh.box(rank), src.data() + w0, cnt * 4);
        if (!sh.barrier(err)) return 3;
        for (int r = 0; r < P; ++r) {
            const uint32_t* b = reinterpret_cast<const uint32_t*>(sh.box(r));
            for (size_t i = 0; i < cnt; i += 97) if (b[i] != (uint32_t)(r << 20) + (uint32_t)(w0 + i)) ++bad;
        }
        if (!sh.barrier(err)) return 3;
    }
    sh.timeout_s = 10;
    sh.close_link();
    return bad ? 1 : 0;
}

int main(int argc, char** argv) {
    const int P = argc > 1 ? atoi(argv[1]) : 3;
    unsigned char id[128];
    for (int i = 0; i < 128; ++i) id[i] = (unsigned char)(i * 37 + getpid());
    // (mailboxes of 64 KiB: the box size is an argument of open() since the library stopped reading the environment)
    {
        // what a crashed run with the same id leaves behind: a complete-looking segment that all its P ranks had attached to.  The ranks
        // below must not settle on it (rank 0 replaces it; a rank that mapped the old one sees the name lead elsewhere and looks again)
        const std::string nm = ShmLink::name_of(id);
        const size_t bytes = 4096 + (size_t)P * (((size_t)1 << 20) + 65536);
        const int fd = shm_open(nm.c_str(), O_CREAT | O_RDWR, 0600);
        if (fd >= 0 && ftruncate(fd, (off_t)bytes) == 0) {
            void* m = mmap(nullptr, bytes, PROT_READ | PROT_WRITE, MAP_SHARED, fd, 0);
            if (m != MAP_FAILED) {
                ShmLink::Header* h = static_cast<ShmLink::Header*>(m);
                h->arrived.store(1); h->sense.store(1); h->attached.store((uint32_t)P);
                h->nranks = (uint32_t)P; h->slot_bytes = (size_t)1 << 20; h->box_bytes = 65536;
                h->magic.store(ShmLink::MAGIC);
                munmap(m, bytes);
            }
        }
        if (fd >= 0) close(fd);
    }
    std::vector<pid_t> kids;
    for (int r = 1; r < P; ++r) {
        pid_t p = fork();
        if (p == 0) _exit(run_rank(r, P, id));
        kids.push_back(p);
    }
    int rc = run_rank(0, P, id);
    for (pid_t p : kids) { int st = 0; waitpid(p, &st, 0); if (!WIFEXITED(st) || WEXITSTATUS(st) != 0) rc = rc ? rc : 10 + WEXITSTATUS(st); }
    // a peer that never arrives: the barrier gives up instead of hanging
    if (rc == 0) {
        unsigned char id2[128];
        for (int i = 0; i < 128; ++i) id2[i] = (unsigned char)(i * 11 + 3);
        ShmLink lone; std::string err;
        lone.timeout_s = 0.3;
        if (lone.open(0, 2, id2, err, 65536)) rc = 20;            // rank 1 is missing: open()'s first barrier must time out
        if (lone.base) { munmap(lone.base, lone.bytes); shm_unlink(lone.name.c_str()); }
    }
    printf(rc == 0 ? "ok\n" : "FAILED %d\n", rc);
    return rc;
}
