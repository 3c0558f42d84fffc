"""One rank of a process-per-rank communicator (psacx_multi_create_rank), run as its own process by the tests:
  python multi_rank_proc.py <rank> <nranks> <device> <uid hex> <kind> <n> <seed> <bits> <out dir> [check] [ansv]
Builds its block of the text, constructs SA / ISA / LCP with the other ranks, writes the blocks to <out dir>/r<rank>_*.npy
together with the wire counters, and optionally runs the distributed checker and the distributed ANSV over the LCP blocks.
psac's own deployment is this shape: one MPI rank per block (src/psac.cpp:85-93)."""
import ctypes as C
import json
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(HERE))
sys.path.insert(0, HERE)
import inputs  # noqa: E402


def make_text(kind, n, seed):
    if kind == "dna":
        return inputs.dna(n, seed)
    if kind == "tandem":
        return inputs.tandem(n, 256, inputs.dna(256, seed))
    if kind == "ascii":
        return inputs.ascii128(n, seed)
    if kind == "single":
        return np.full(n, 65, np.uint8)
    raise SystemExit("unknown kind " + kind)


def main():
    rank, P, dev = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3])
    uid = bytes.fromhex(sys.argv[4])
    kind, n, seed, bits, out = sys.argv[5], int(sys.argv[6]), int(sys.argv[7]), int(sys.argv[8]), sys.argv[9]
    extras = sys.argv[10:]
    import psac_amd
    text = make_text(kind, n, seed)
    sizes = [n // P + (1 if r < n % P else 0) for r in range(P)]
    off = sum(sizes[:rank])
    m = sizes[rank]
    dt = np.uint32 if bits == 32 else np.uint64
    w = bits // 8
    mg = psac_amd.MultiContext.for_rank(rank, P, dev, uid)
    lib = mg._lib
    ctx = mg.rank_ctx(0)

    def alloc(nb):
        p = C.c_void_p()
        assert lib.psacx_dev_alloc(ctx, C.byref(p), max(nb, 16)) == 0
        return p.value

    def up(a):
        a = np.ascontiguousarray(a)
        p = alloc(a.nbytes)
        if a.nbytes:
            assert lib.psacx_copy_h2d(ctx, C.c_void_p(p), a.ctypes.data_as(C.c_void_p), a.nbytes) == 0
        return p

    def down(p, count, dtype):
        a = np.empty(count, dtype)
        if count:
            assert lib.psacx_copy_d2h(ctx, a.ctypes.data_as(C.c_void_p), C.c_void_p(p), a.nbytes) == 0
        return a

    d_text = up(text[off:off + m])
    d_sa, d_isa, d_lcp = alloc(m * w), alloc(m * w), alloc(m * w)
    st, sent, nex, nga = mg.construct_device([d_text], [m], [d_sa], [d_isa], [d_lcp], bits)
    np.save(os.path.join(out, "r%d_sa.npy" % rank), down(d_sa, m, dt))
    np.save(os.path.join(out, "r%d_isa.npy" % rank), down(d_isa, m, dt))
    np.save(os.path.join(out, "r%d_lcp.npy" % rank), down(d_lcp, m, dt))
    info = {"rank": rank, "nranks": mg.nranks, "nlocal": mg.nlocal, "transport": mg.transport, "bytes_sent": sent,
            "exchanges": nex, "gathers": nga, "wire": mg.wire(), "phases": mg.phases(),
            "rounds": [(int(r.h), int(r.unfinished_buckets), int(r.unfinished_elements)) for r in st.rounds[:st.n_rounds]]}
    if "check" in extras:
        info["check"] = mg.check_device([d_text], [m], [d_sa], [d_isa], [d_lcp], bits)
    if "ansv" in extras:
        d_l, d_r = alloc(m * 8), alloc(m * 8)
        mg.ansv_device([d_lcp], [m], [d_l], [d_r], bits, left_type=2, right_type=0, nonsv=n)
        np.save(os.path.join(out, "r%d_left.npy" % rank), down(d_l, m, np.uint64))
        np.save(os.path.join(out, "r%d_right.npy" % rank), down(d_r, m, np.uint64))
    with open(os.path.join(out, "r%d.json" % rank), "w") as f:
        json.dump(info, f)
    mg.close()


if __name__ == "__main__":
    main()
