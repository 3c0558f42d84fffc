"""psac over one or several MI355X: `python -m psac_amd (-f <file> | -r <size>) [-s seed] [-o base] [-l] [-c]`.

The command line of /root/reference/src/psac.cpp:56-153 for the block-distributed path: start it once
per GPU with torchrun (`python -m torch.distributed.run --nproc-per-node N --master-addr 127.0.0.1
-m psac_amd ...`), the way the reference is started with `mpirun -np N psac ...`.

* input: every rank reads its block of the file (mxx::file_block_decompose, src/psac.cpp:85), or draws
  size / N random DNA characters with seed * rank (src/psac.cpp:88, alphabet.hpp:32-45, glibc rand);
* output: `<base>.sa64` (and `.lcp64` with -l), raw little-endian uint64 in global order, every rank
  writing its block (mxx::write_ordered, src/psac.cpp:123-128);
* `-c`: the arrays are gathered on rank 0 and verified there by the device checker
  (check_suffix_array.hpp:56-126); prints the reference's `[SUCCESS]` / `[ERROR]` lines.
With a single process the single-GPU engine is used; `-t` (suffix tree) is only available in the C++
`psac` binary.  Extra flag: `--index {32,64,auto}` (files stay uint64).
"""
import argparse
import ctypes
import os
import sys
import time

import numpy as np


def _rand_dna(size, seed):
    """alphabet.hpp:32-45: srand(1337 * seed); "ACGT"[rand() % 4] with the C library's generator
    (filled by psacx_rand_dna in libpsacx.so: one native loop, the same glibc sequence as the reference)."""
    from psac_amd import _lib
    out = np.empty(size, np.uint8)
    rc = _lib.load().psacx_rand_dna(out.ctypes.data_as(ctypes.c_void_p), size, int(seed))
    if rc != 0:
        raise RuntimeError("psacx_rand_dna failed: %d" % rc)
    return out


def _blk(n, P, r):
    m = n // P + (1 if r < n % P else 0)
    off = r * (n // P) + min(r, n % P)
    return off, m


def _write_block(path, arr_u64, off_elems, total_elems, rank, barrier):
    if rank == 0:
        with open(path, "wb") as f:
            f.truncate(total_elems * 8)
    barrier()
    with open(path, "r+b") as f:
        f.seek(off_elems * 8)
        f.write(arr_u64.tobytes())
    barrier()


def main(argv=None):
    ap = argparse.ArgumentParser(prog="python -m psac_amd",
                                 description="Parallel distributed suffix array and LCP construction (MI355X engine).")
    g = ap.add_mutually_exclusive_group(required=True)
    g.add_argument("-f", "--file")
    g.add_argument("-r", "--random", type=int)
    ap.add_argument("-o", "--outfile", default="")
    ap.add_argument("-s", "--seed", type=int, default=0)
    ap.add_argument("-l", "--lcp", action="store_true")
    ap.add_argument("-c", "--check", action="store_true")
    ap.add_argument("--index", default="auto", choices=("32", "64", "auto"))
    a = ap.parse_args(argv)

    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    import torch
    import psac_amd

    if a.file:
        n = os.path.getsize(a.file)
        off, m = _blk(n, world, rank)
        with open(a.file, "rb") as f:
            f.seek(off)
            text = np.frombuffer(f.read(m), np.uint8).copy()
    else:
        m = a.random // world                       # src/psac.cpp:88
        n = m * world
        off = rank * m
        text = _rand_dna(m, a.seed * rank)
    if n == 0:
        sys.stderr.write("error: empty input\n")
        return 1
    bits = 32 if (a.index == "32" or (a.index == "auto" and n < 0xFFFFFFFE)) else 64
    udt = np.uint32 if bits == 32 else np.uint64

    single = world == 1 and not os.environ.get("PSACX_CLI_FORCE_DIST")
    if single:
        sa = psac_amd.SuffixArray(index_bits=bits, lcp=a.lcp, ctx=psac_amd.Context(local_rank), log=sys.stderr)
        t0 = time.perf_counter()
        sa.construct(text)
        sys.stderr.write("PSAC time: %g ms\n" % ((time.perf_counter() - t0) * 1e3))
        SA, ISA, LCP = sa.local_SA, sa.local_B, (sa.local_LCP if a.lcp else None)
        barrier = lambda: None
        gathered = (text, SA, ISA, LCP)
    else:
        # one process per GPU (psac's deployment: one MPI rank per block): the C++ multi-GPU engine behind
        # psacx_multi_*; torch.distributed only carries the communicator id and the barriers
        import ctypes as C
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        torch.cuda.set_device(local_rank)
        dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))
        box = [psac_amd.unique_id() if rank == 0 else None]
        dist.broadcast_object_list(box, src=0)
        mg = psac_amd.MultiContext.for_rank(rank, world, local_rank, box[0])
        lib, rctx = mg._lib, mg.rank_ctx(0)
        w = bits // 8

        def dalloc(nbytes):
            p = C.c_void_p()
            if lib.psacx_dev_alloc(rctx, C.byref(p), max(int(nbytes), 1)) != 0:
                raise RuntimeError("device allocation failed")
            return p.value

        def barrier():
            dist.barrier()
            torch.cuda.synchronize()
        d_text, d_sa, d_isa = dalloc(m), dalloc(m * w), dalloc(m * w)
        d_lcp = dalloc(m * w) if a.lcp else None
        bufs = [d_text, d_sa, d_isa] + ([d_lcp] if a.lcp else [])

        def ok(rcode, what):
            if rcode != 0:
                raise RuntimeError("%s failed on rank %d (status %d)" % (what, rank, rcode))
        ok(lib.psacx_copy_h2d(rctx, C.c_void_p(d_text), text.ctypes.data_as(C.c_void_p), m), "copy of the text block to the device")
        barrier()
        t0 = time.perf_counter()
        st = mg.construct_device([d_text], [m], [d_sa], [d_isa], [d_lcp] if a.lcp else None, bits)[0]
        barrier()
        if rank == 0:
            for r in st.rounds[:st.n_rounds]:
                sys.stderr.write("iteration %d: unfinished buckets = %d, unfinished elements = %d\n" % (r.h, r.unfinished_buckets, r.unfinished_elements))
            sys.stderr.write("PSAC time: %g ms\n" % ((time.perf_counter() - t0) * 1e3))
        SA = np.empty(m, udt); LCP = np.empty(m, udt) if a.lcp else None
        if rank == 0 and mg.memory()[2]:
            sys.stderr.write("note: %d refinement rounds ran in slabs (reduced-memory layout); their counters may run ahead of psac's log\n" % mg.memory()[2])
        ok(lib.psacx_copy_d2h(rctx, SA.ctypes.data_as(C.c_void_p), C.c_void_p(d_sa), m * w), "copy of the SA block to the host")
        if a.lcp:
            ok(lib.psacx_copy_d2h(rctx, LCP.ctypes.data_as(C.c_void_p), C.c_void_p(d_lcp), m * w), "copy of the LCP block to the host")
        rc = 0
        if a.check:
            # the distributed checker (d_check_sa, check_suffix_array.hpp:207-267, plus the LCP recurrence): nothing is
            # gathered on one rank, so it works at any size the construction itself works at
            err = mg.check_device([d_text], [m], [d_sa], [d_isa], [d_lcp] if a.lcp else None, bits)
            if any(err):
                if rank == 0:
                    sys.stderr.write("[ERROR] Test unsuccessful %s\n" % err)
                rc = 1
            elif rank == 0:
                sys.stderr.write("[SUCCESS] Suffix Array%s are correct\n" % (" and LCP" if a.lcp else ""))
        for p in bufs:
            lib.psacx_dev_free(rctx, C.c_void_p(p))

    if single:
        rc = 0
    if a.check and single:
        gt, gsa, gisa, glcp = gathered
        ctx = psac_amd.Context(local_rank)
        w = bits // 8
        d_t, d_s, d_i = ctx.alloc(n), ctx.alloc(n * w), ctx.alloc(n * w)
        d_l = ctx.alloc(n * w) if a.lcp else 0
        ctx.h2d(d_t, gt); ctx.h2d(d_s, gsa); ctx.h2d(d_i, gisa)
        if a.lcp:
            ctx.h2d(d_l, glcp)
        err = psac_amd.check_device(ctx, d_t, n, d_s, d_i, d_l, bits)
        if any(err):
            sys.stderr.write("[ERROR] Test unsuccessful\n")
            rc = 1
        else:
            sys.stderr.write("[SUCCESS] Suffix Array%s are correct\n" % (" and LCP" if a.lcp else ""))
    if a.outfile:
        _write_block(a.outfile + ".sa64", SA.astype(np.uint64), off, n, rank, barrier)
        if a.lcp:
            _write_block(a.outfile + ".lcp64", LCP.astype(np.uint64), off, n, rank, barrier)
    if not single:
        mg.close()
        dist.destroy_process_group()
    return rc


if __name__ == "__main__":
    sys.exit(main())
