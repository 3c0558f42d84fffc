"""The multi-GPU construction behind the C ABI (psacx_multi_*, psac_amd/csrc/multi.hpp): C++ host code, HIP step
kernels, exchanges on a second stream.  A test box has ONE GPU, so the ranks here share device 0 (dev_ids = [0] * P):
the choreography, the partitioning, the sample sort and every exchange are the ones a node with P GPUs runs, only the
transport is device-to-device copies instead of RCCL (which refuses two ranks on one device).  With one rank and a
unique id the RCCL path itself (dlopen, ncclCommInitRank, the group calls) is exercised at world size 1.
Bit-exact against the oracle and, independently, libdivsufsort + Kasai."""
import numpy as np
import pytest

import inputs
import oracle_lib as O

pytestmark = pytest.mark.gpu


def multi(P):
    import psac_amd
    return psac_amd.MultiContext([0] * P)


def same(mg, text, bits, k=0, lcp=True):
    SA, ISA, LCP, rounds = mg.construct(text, index_bits=bits, lcp=lcp, k=k)
    return SA, ISA, LCP, rounds


@pytest.mark.parametrize("P", [1, 2, 3, 4, 7])
def test_multi_matches_oracle(P):
    mg = multi(P)
    try:
        for bits in (32, 64):
            text = O.rand_dna(60011, 7)
            SA, ISA, LCP, rounds = same(mg, text, bits)
            ref = O.construct(text, bits=bits)
            assert np.array_equal(SA, ref["SA"]) and np.array_equal(ISA, ref["ISA"]) and np.array_equal(LCP, ref["LCP"])
            assert rounds == [(h, b, e) for h, b, e, _ in ref["trace"]]
        # deep rounds (tandem repeat): every round has range minima that cross rank boundaries
        text = inputs.tandem(40000, 256, O.rand_dna(256, 3))
        SA, ISA, LCP, rounds = same(mg, text, 32)
        ref = O.construct(text, bits=32)
        assert np.array_equal(SA, ref["SA"]) and np.array_equal(ISA, ref["ISA"]) and np.array_equal(LCP, ref["LCP"])
        assert rounds == [(h, b, e) for h, b, e, _ in ref["trace"]]
        # forced bucket refinement (k = 3), no LCP
        text = O.rand_dna(30011, 23)
        SA, ISA, LCP, _ = same(mg, text, 64, k=3, lcp=False)
        assert LCP is None and np.array_equal(SA, O.naive_sa(text, 64))
        assert np.array_equal(ISA[SA.astype(np.int64)], np.arange(text.size, dtype=np.uint64))
        # heavy ties: a single symbol, and a text whose length is not a multiple of P
        text = np.full(5003, 65, np.uint8)
        SA, ISA, LCP, _ = same(mg, text, 32)
        ref = O.construct(text, bits=32)
        assert np.array_equal(SA, ref["SA"]) and np.array_equal(LCP, ref["LCP"])
    finally:
        mg.close()


@pytest.mark.parametrize("P", [2, 3, 8])
def test_multi_refinement_sort_forms(P, monkeypatch):
    # refinement rounds on p ranks: the buckets that reach over a rank boundary are sorted across the ranks, all others where they lie
    # (multi.hpp: refine_sort; two-word local records with 64-bit words, three-word ones with 32-bit words); PSACX_MULTI_GLOBAL_REFINE_SORT=1
    # keeps the sort of all records across the ranks.  A tandem repeat (buckets longer than a block early on, every rank boundary inside a
    # bucket), repeated reads with mutations (many small buckets, few of them on a boundary) and one symbol (one bucket over all ranks).
    texts = [inputs.tandem(150001, 512, O.rand_dna(512, 3)), inputs.mutated(120007, 2048, 5), np.full(20011, 71, np.uint8)]
    for env in (None, "1"):
        if env: monkeypatch.setenv("PSACX_MULTI_GLOBAL_REFINE_SORT", env)
        mg = multi(P)
        try:
            for text in texts:
                for bits in (64, 32):
                    SA, ISA, LCP, rounds = same(mg, text, bits)
                    ref = O.construct(text, bits=bits)
                    assert np.array_equal(SA, ref["SA"]) and np.array_equal(ISA, ref["ISA"]) and np.array_equal(LCP, ref["LCP"])
                    assert rounds == [(h, b, e) for h, b, e, _ in ref["trace"]]
        finally:
            mg.close()


def test_multi_block_decomposition_is_enforced():
    # suffix_array.hpp:226-227: blocks that do not follow mxx::blk_dist are refused
    import ctypes as C
    import psac_amd
    mg = multi(2)
    try:
        c0, c1 = mg.rank_ctx(0), mg.rank_ctx(1)
        lib = mg._lib
        ptrs = []
        def alloc(ctx, nbytes):
            p = C.c_void_p()
            assert lib.psacx_dev_alloc(ctx, C.byref(p), nbytes) == 0
            ptrs.append((ctx, p))
            return p.value
        m = [10, 30]
        t = [alloc(c0, 64), alloc(c1, 64)]
        out = [[alloc(c, 64 * 8) for c in (c0, c1)] for _ in range(3)]
        with pytest.raises(psac_amd.PsacxError) as e:
            mg.construct_device(t, m, out[0], out[1], out[2], 64)
        assert "equally block decomposed" in str(e.value)
        for ctx, p in ptrs:
            lib.psacx_dev_free(ctx, p)
    finally:
        mg.close()


def test_multi_larger_and_low_entropy():
    mg = multi(3)
    try:
        text = inputs.dna((1 << 22) + 1234, 9)
        SA, ISA, LCP, _ = same(mg, text, 32)
        assert O.check_sa(text, SA, ISA) == 0
        assert np.array_equal(O.kasai(text, SA, ISA), LCP)
        rng = np.random.RandomState(5)
        p = 0.5 ** np.arange(1, 21); p /= p.sum()
        text = (97 + rng.choice(20, size=(1 << 20) + 77, p=p)).astype(np.uint8)
        SA, ISA, LCP, rounds = same(mg, text, 32)
        ref = O.construct(text, bits=32)
        assert np.array_equal(SA, ref["SA"]) and np.array_equal(ISA, ref["ISA"]) and np.array_equal(LCP, ref["LCP"])
        assert rounds == [(h, b, e) for h, b, e, _ in ref["trace"]]
    finally:
        mg.close()


def test_multi_twins_of_the_eight_gpu_configs():
    # BASELINE.json configs[3] / 256: 2^26 random DNA over 8 ranks, uint64; configs[4] / 256: 2^27 characters of a
    # period-1024 tandem repeat of DNA(1024, 3) over 8 ranks, uint64 (deep prefix doubling: ~23 rounds)
    mg = multi(8)
    try:
        text = inputs.dna(1 << 26, 1)
        SA, ISA, LCP, _ = same(mg, text, 64)
        rSA, rLCP = O.construct_all_cores(text, bits=64)
        assert np.array_equal(SA, rSA) and np.array_equal(LCP, rLCP)
        assert np.array_equal(ISA[SA.astype(np.int64)], np.arange(text.size, dtype=np.uint64))
        if O.have_divsufsort():
            assert np.array_equal(SA, O.divsufsort(text, 64))
        del rSA, rLCP
        text = inputs.tandem(1 << 27, 1024, inputs.dna(1024, 3))
        SA, ISA, LCP, rounds = same(mg, text, 64)
        rSA, rLCP = O.reference_sa_lcp_cached("tandem_1024_3", text, bits=64)
        assert np.array_equal(SA, rSA) and np.array_equal(LCP, rLCP)
        assert np.array_equal(ISA[SA.astype(np.int64)], np.arange(text.size, dtype=np.uint64))
        assert [r[0] for r in rounds] == [21 << i for i in range(len(rounds))] and len(rounds) >= 20
        st, sent, ex, ga = mg.stats()
        assert sent > 0 and ex > 0
    finally:
        mg.close()


def test_multi_rccl_path_at_world_size_one():
    # one process per GPU with a communicator built from a unique id (what bench.py --gpus N does under torchrun)
    import psac_amd
    uid = psac_amd.unique_id()
    assert len(uid) == 128
    mg = psac_amd.MultiContext.for_rank(0, 1, 0, uid)
    try:
        assert mg.nranks == 1 and mg.nlocal == 1
        text = O.rand_dna(100003, 5)
        SA, ISA, LCP, _ = mg.construct(text, index_bits=32)
        ref = O.construct(text, bits=32)
        assert np.array_equal(SA, ref["SA"]) and np.array_equal(ISA, ref["ISA"]) and np.array_equal(LCP, ref["LCP"])
    finally:
        mg.close()


def _blocks_on_device(mg, text, bits, P):
    """Uploads the blocks of `text` to the ranks of mg and constructs; returns the device addresses and a free()."""
    import ctypes as C
    lib = mg._lib
    n = text.size
    w = bits // 8
    sizes = [n // P + (1 if r < n % P else 0) for r in range(P)]
    offs = np.concatenate([[0], np.cumsum(sizes)]).astype(np.int64)
    held = []

    def alloc(ctx, nbytes):
        p = C.c_void_p()
        assert lib.psacx_dev_alloc(ctx, C.byref(p), max(nbytes, 1)) == 0
        held.append((ctx, p))
        return p.value
    d = dict(text=[], sa=[], isa=[], lcp=[])
    for r in range(P):
        ctx = mg.rank_ctx(r)
        d["text"].append(alloc(ctx, sizes[r]))
        blk = np.ascontiguousarray(text[offs[r]:offs[r + 1]])
        assert lib.psacx_copy_h2d(ctx, C.c_void_p(d["text"][r]), blk.ctypes.data_as(C.c_void_p), sizes[r]) == 0
        for key in ("sa", "isa", "lcp"):
            d[key].append(alloc(ctx, sizes[r] * w))

    def free():
        for ctx, p in held:
            lib.psacx_dev_free(ctx, p)
    return d, sizes, free


@pytest.mark.parametrize("chunks", [0, 7])
def test_multi_distributed_checker(chunks, monkeypatch):
    # d_check_sa + the LCP recurrence over block-distributed results (nothing gathered on one rank); a repetitive text
    # has range minima that span ranks; every kind of corruption must be counted.  chunks = 7: every block verified in
    # seven pieces of consecutive SA positions (what the checker does by itself when a block is a large share of the device)
    import ctypes as C
    P, bits = 4, 32
    if chunks:
        monkeypatch.setenv("PSACX_MULTI_CHECK_CHUNKS", str(chunks))
    mg = multi(P)
    try:
        for text in (inputs.dna(300007, 4), inputs.tandem(120000, 512, O.rand_dna(512, 2)), np.full(9001, 66, np.uint8)):
            d, sizes, free = _blocks_on_device(mg, text, bits, P)
            mg.construct_device(d["text"], sizes, d["sa"], d["isa"], d["lcp"], bits)
            assert mg.check_device(d["text"], sizes, d["sa"], d["isa"], d["lcp"], bits) == [0, 0, 0, 0]
            assert mg.check_device(d["text"], sizes, d["sa"], d["isa"], None, bits)[:2] == [0, 0]
            lib = mg._lib
            # one wrong LCP entry on rank 2, then two swapped SA entries on rank 1
            bad = np.array([12345], np.uint32)
            keep = np.empty(1, np.uint32)
            at = d["lcp"][2] + 100 * 4
            lib.psacx_copy_d2h(mg.rank_ctx(2), keep.ctypes.data_as(C.c_void_p), C.c_void_p(at), 4)
            lib.psacx_copy_h2d(mg.rank_ctx(2), C.c_void_p(at), bad.ctypes.data_as(C.c_void_p), 4)
            err = mg.check_device(d["text"], sizes, d["sa"], d["isa"], d["lcp"], bits)
            assert err[0] == 0 and err[1] == 0 and err[2] >= 1
            lib.psacx_copy_h2d(mg.rank_ctx(2), C.c_void_p(at), keep.ctypes.data_as(C.c_void_p), 4)
            two = np.empty(2, np.uint32)
            at = d["sa"][1] + 50 * 4
            lib.psacx_copy_d2h(mg.rank_ctx(1), two.ctypes.data_as(C.c_void_p), C.c_void_p(at), 8)
            sw = two[::-1].copy()
            lib.psacx_copy_h2d(mg.rank_ctx(1), C.c_void_p(at), sw.ctypes.data_as(C.c_void_p), 8)
            err = mg.check_device(d["text"], sizes, d["sa"], d["isa"], d["lcp"], bits)
            assert err[0] >= 2                      # ISA is no longer the inverse at the two positions
            free()
    finally:
        mg.close()


def test_psac_cli_and_cpp_header_on_several_ranks(tmp_path):
    # psac --gpus N (here N ranks on device 0) and suffix_array<> with a multi-device communicator: same outputs as one rank
    import os
    import subprocess
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    psac = os.path.join(root, "psac_amd", "bin", "psac")
    if not os.path.exists(psac):
        pytest.skip("CLI not built")
    text = O.rand_dna(250003, 9)
    f = tmp_path / "t.txt"
    f.write_bytes(bytes(text))
    r = subprocess.run([psac, "-f", str(f), "-l", "-c", "-o", str(tmp_path / "m"), "--gpus-on-device", "0,3"], capture_output=True, text=True)
    assert r.returncode == 0 and "[SUCCESS]" in r.stderr and "PSAC time:" in r.stderr, r.stderr
    ref = O.construct(text, bits=32)
    assert np.array_equal(np.fromfile(str(tmp_path / "m.sa64"), np.uint64), ref["SA"].astype(np.uint64))
    assert np.array_equal(np.fromfile(str(tmp_path / "m.lcp64"), np.uint64), ref["LCP"].astype(np.uint64))
    # psac -t on several ranks (src/psac.cpp:96-114): the same number of suffix-tree edges as on one
    edges = []
    for extra in ([], ["--gpus-on-device", "0,3"]):
        r = subprocess.run([psac, "-f", str(f), "-t"] + extra, capture_output=True, text=True)
        assert r.returncode == 0 and "ST time:" in r.stderr, r.stderr
        edges.append([ln for ln in r.stderr.splitlines() if ln.startswith("ST edges:")])
    assert edges[0] and edges[0] == edges[1]
    src = tmp_path / "p.cpp"
    src.write_text(r'''
#include "suffix_array.hpp"
#include <cstdio>
int main() {
    std::string s;
    srand(1337 * 5);
    for (int i = 0; i < 70001; ++i) s.push_back("ACGT"[rand() % 4]);
    suffix_array<char, uint64_t, true> one((psacx::comm(0)));
    one.verbose = false;
    one.construct(s.begin(), s.end());
    suffix_array<char, uint64_t, true> many((psacx::comm(std::vector<int>(4, 0))));
    many.verbose = false;
    many.construct(s.begin(), s.end());
    if (many.p != 4 || many.n != s.size()) return 2;
    if (one.local_SA != many.local_SA || one.local_B != many.local_B || one.local_LCP != many.local_LCP) return 3;
    many.construct(s.begin(), s.begin() + 50000);          // repeated calls on one object (test/test_psac.cpp:148-170)
    one.construct(s.begin(), s.begin() + 50000);
    if (one.local_SA != many.local_SA || one.local_LCP != many.local_LCP) return 4;
    // left-branching characters on four ranks (suffix_array<char, index_t, true, true>, suffix_array.hpp:211-212)
    suffix_array<char, uint32_t, true, true> lc1((psacx::comm(0)));
    lc1.verbose = false;
    lc1.construct(s.begin(), s.end());
    suffix_array<char, uint32_t, true, true> lc4((psacx::comm(std::vector<int>(4, 0))));
    lc4.verbose = false;
    lc4.construct(s.begin(), s.end());
    if (lc1.local_Lc.size() != s.size() || lc1.local_Lc != lc4.local_Lc || lc1.local_SA != lc4.local_SA) return 5;
    for (size_t i = 1; i < s.size(); ++i) {
        const size_t p = (size_t)lc4.local_SA[i - 1] + (size_t)lc4.local_LCP[i];
        if (lc4.local_Lc[i] != (p < s.size() ? s[p] : '\0')) return 6;
    }
    // the suffix-tree node table built by the four ranks of the suffix array's communicator (suffix_tree.hpp:413-499)
    std::vector<size_t> t1 = construct_suffix_tree(one, s.begin(), s.begin() + 50000, psacx::comm(0));
    std::vector<size_t> t4 = construct_suffix_tree(many, s.begin(), s.begin() + 50000, psacx::comm(std::vector<int>(4, 0)));
    if (t1.size() != 5 * 50000u || t1 != t4) return 7;
    std::puts("ok");
    return 0;
}
''')
    exe = str(tmp_path / "p")
    lib = os.path.join(root, "psac_amd", "lib")
    subprocess.check_call(["g++", "-std=c++11", "-O2", "-I" + os.path.join(root, "include"), "-o", exe, str(src), "-L" + lib, "-lpsacx",
                           "-Wl,-rpath," + lib, "-Wl,-rpath,/opt/rocm/lib"])
    r = subprocess.run([exe], capture_output=True, text=True)
    assert r.returncode == 0 and "ok" in r.stdout, (r.returncode, r.stderr)


def test_bench_and_cli_process_per_gpu_path(tmp_path):
    # what the driver launches for N > 1, at world size 1 on this box: torchrun -> bench.py --gpus ... -> psacx_multi_create_rank
    import json
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, PSACX_BENCH_FORCE_DIST="1", MASTER_ADDR="127.0.0.1")
    r = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "1", "--master-addr", "127.0.0.1",
                        "--master-port", "29577", os.path.join(root, "bench.py"), "--gpus", "1", "--steps", "2", "--warmup", "1", "--size", str(1 << 22)],
                       capture_output=True, text=True, env=env, cwd=root, timeout=600)
    assert r.returncode == 0, r.stderr[-3000:]
    line = [l for l in r.stdout.splitlines() if l.startswith("{")][-1]
    out = json.loads(line)
    assert out["n_gpus"] == 1 and out["value"] > 0 and "roofline" in out and out["config"]["rounds"] >= 1
    # what makes the N > 1 line readable: the one-GPU engine on the same block and the efficiency against it, what RCCL saw, the memory
    # peak per rank, the piece size of the wire, the forms the construction took -- and, with the wire forced and the one-word first round
    # at this size, a roofline entry quoted on the bucket passes (16 or 24 bytes per record) with every send matched by a receive
    env2 = dict(env, PSACX_MULTI_FORCE_WIRE="1", PSACX_MULTI_TWO_WORD="1", PSACX_MULTI_ONE_WORD="1")
    r = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "1", "--master-addr", "127.0.0.1",
                        "--master-port", "29579", os.path.join(root, "bench.py"), "--gpus", "1", "--steps", "2", "--warmup", "1", "--size", str(1 << 24), "--index", "64"],
                       capture_output=True, text=True, env=env2, cwd=root, timeout=600)
    assert r.returncode == 0, r.stderr[-3000:]
    out = json.loads([l for l in r.stdout.splitlines() if l.startswith("{")][-1])
    assert out["config"]["one_gpu_engine_same_block"]["ms"] > 0 and 0 < out["weak_scaling_efficiency"] < 2
    ex = out["exchange"]
    assert ex["ranks_seen_by_rccl"] == [1] and ex["transport"] == "rccl" and ex["wire_piece_bytes"] == 1 << 28
    assert ex["forms_per_rank"][0]["one_word"] and ex["nccl_calls_last_step_rank0"]["sends"] == ex["nccl_calls_last_step_rank0"]["recvs"] > 0
    assert out["config"]["layout"]["engine_words_per_char_at_peak_per_rank"][0] > 0
    assert 16 <= out["roofline"]["bytes_per_record_per_pass"] <= 24 and out["roofline"]["records_per_launch"] > 0
    env = dict(os.environ, PSACX_CLI_FORCE_DIST="1", MASTER_ADDR="127.0.0.1")
    r = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "1", "--master-addr", "127.0.0.1",
                        "--master-port", "29578", "-m", "psac_amd", "-r", "300000", "-s", "2", "-l", "-c"], capture_output=True, text=True, env=env,
                       cwd=root, timeout=600)
    assert r.returncode == 0 and "[SUCCESS]" in r.stderr, r.stderr[-3000:]


@pytest.mark.parametrize("P", [1, 2, 3, 7])
def test_multi_left_branching_chars(P):
    # psacx_multi_left_chars_dev_*: local_Lc of suffix_array<char, index_t, true, true> on p ranks (suffix_array.hpp:211-212,
    # :1365-1383, par_rmq.hpp:334-481) -- against the oracle, which carries Lc through the leftmost range minima as the
    # reference does, and against the definition Lc[i] = S[SA[i-1] + LCP[i]] (desa.hpp:262-264)
    import ctypes as C
    mg = multi(P)
    lib = mg._lib
    cases = [(O.as_text("mississippi"), 64), (O.rand_dna(130370, 7), 32), (inputs.tandem(100000, 1024, inputs.dna(1024, 5)), 64),
             (inputs.ascii128(60000, 9), 32), (O.as_text("aaaaaaaaaaaaaaaa"), 32), (inputs.cyclic(39999, "abc"), 64)]
    try:
        for text, bits in cases:
            n = text.size
            if n < P:
                continue
            w = bits // 8
            udt = np.uint32 if bits == 32 else np.uint64
            sizes = [n // P + (1 if r < n % P else 0) for r in range(P)]
            offs = np.concatenate([[0], np.cumsum(sizes)]).astype(np.int64)
            held = []

            def alloc(ctx, nbytes):
                p = C.c_void_p()
                assert lib.psacx_dev_alloc(ctx, C.byref(p), max(nbytes, 1)) == 0
                held.append((ctx, p))
                return p.value
            d_t, d_sa, d_isa, d_lcp, d_lc = [], [], [], [], []
            for r in range(P):
                ctx = mg.rank_ctx(r)
                d_t.append(alloc(ctx, sizes[r])); d_lc.append(alloc(ctx, sizes[r]))
                for lst in (d_sa, d_isa, d_lcp):
                    lst.append(alloc(ctx, sizes[r] * w))
                blk = np.ascontiguousarray(text[offs[r]:offs[r + 1]])
                assert lib.psacx_copy_h2d(ctx, C.c_void_p(d_t[r]), blk.ctypes.data_as(C.c_void_p), sizes[r]) == 0
            mg.construct_device(d_t, sizes, d_sa, d_isa, d_lcp, bits)
            mg.left_chars_device(d_t, sizes, d_sa, d_lcp, d_lc, bits)
            SA = np.empty(n, udt); LCP = np.empty(n, udt); Lc = np.empty(n, np.uint8)
            for r in range(P):
                ctx = mg.rank_ctx(r)
                for dst, src, ww in ((SA, d_sa, w), (LCP, d_lcp, w), (Lc, d_lc, 1)):
                    part = np.empty(sizes[r], dst.dtype)
                    assert lib.psacx_copy_d2h(ctx, part.ctypes.data_as(C.c_void_p), C.c_void_p(src[r]), sizes[r] * ww) == 0
                    dst[offs[r]:offs[r + 1]] = part
            ref = O.construct_lc(text, bits=bits)
            assert np.array_equal(SA, ref["SA"]) and np.array_equal(LCP, ref["LCP"])
            assert np.array_equal(Lc, ref["Lc"])
            assert np.array_equal(Lc, O.left_chars_by_definition(text, SA, LCP))
            for ctx, p in held:
                lib.psacx_dev_free(ctx, p)
    finally:
        mg.close()


def test_multi_distributed_ansv():
    # psacx_multi_ansv_dev_*: ansv<T, left, right, global_indexing> over a block-distributed array, all nine type pairs,
    # against the oracle's restatement of the reference's result contract (ansv.hpp:48-65, ansv_common.hpp:20-22)
    import ctypes as C
    import psac_amd
    rng = np.random.RandomState(8)
    text = inputs.dna(200000, 6)
    ctx1 = psac_amd.Context(0)
    sa = psac_amd.SuffixArray(index_bits=32, lcp=True, ctx=ctx1)
    sa.construct(text)
    lcp = sa.local_LCP.copy()
    ctx1.close()
    cases = [(rng.randint(0, 4, size=5000), 32), (rng.randint(0, 10**6, size=70000), 64), (lcp, 32), (np.zeros(3000), 64),
             (np.arange(5000), 32), (np.arange(9000)[::-1].copy(), 64), (rng.randint(0, 3, size=300001), 32)]
    for P in (1, 2, 4, 7):
        mg = multi(P)
        lib = mg._lib
        try:
            for vals, bits in cases:
                udt = np.uint32 if bits == 32 else np.uint64
                v = np.ascontiguousarray(vals.astype(udt))
                n = v.size
                w = bits // 8
                none = (1 << 64) - 1
                sizes = [n // P + (1 if r < n % P else 0) for r in range(P)]
                offs = np.concatenate([[0], np.cumsum(sizes)]).astype(np.int64)
                held = []

                def alloc(ctx, nbytes):
                    p = C.c_void_p()
                    assert lib.psacx_dev_alloc(ctx, C.byref(p), max(nbytes, 1)) == 0
                    held.append((ctx, p))
                    return p.value
                d_in, d_l, d_r = [], [], []
                for r in range(P):
                    ctx = mg.rank_ctx(r)
                    d_in.append(alloc(ctx, sizes[r] * w)); d_l.append(alloc(ctx, sizes[r] * 8)); d_r.append(alloc(ctx, sizes[r] * 8))
                    blk = np.ascontiguousarray(v[offs[r]:offs[r + 1]])
                    assert lib.psacx_copy_h2d(ctx, C.c_void_p(d_in[r]), blk.ctypes.data_as(C.c_void_p), sizes[r] * w) == 0
                pairs = [(a, b) for a in (0, 1, 2) for b in (0, 1, 2)] if n <= 70000 else [(0, 0), (2, 0), (1, 2)]
                for lt, rt in pairs:
                    mg.ansv_device(d_in, sizes, d_l, d_r, bits, lt, rt, none)
                    Lres = np.empty(n, np.uint64); Rres = np.empty(n, np.uint64)
                    for r in range(P):
                        ctx = mg.rank_ctx(r)
                        lib.psacx_copy_d2h(ctx, Lres[offs[r]:offs[r + 1]].ctypes.data_as(C.c_void_p), C.c_void_p(d_l[r]), sizes[r] * 8)
                        lib.psacx_copy_d2h(ctx, Rres[offs[r]:offs[r + 1]].ctypes.data_as(C.c_void_p), C.c_void_p(d_r[r]), sizes[r] * 8)
                    assert np.array_equal(Lres, O.ansv(v, True, lt, none)), (bits, P, lt, n)
                    assert np.array_equal(Rres, O.ansv(v, False, rt, none)), (bits, P, rt, n)
                for ctx, p in held:
                    lib.psacx_dev_free(ctx, p)
        finally:
            mg.close()


# ---------------------------------------------------------------------------------------------------------------
# reduced-memory layout (psacx_multi_configure): records of the first round in the result arrays + one allocated set,
# chunked SA -> ISA, refinement rounds in slabs of whole buckets
# ---------------------------------------------------------------------------------------------------------------
@pytest.mark.parametrize("P", [1, 2, 3, 8])
def test_multi_reduced_memory_layout_matches_oracle(P):
    mg = multi(P)
    try:
        mg.configure(layout=mg.LAYOUT_REDUCED, slab=3000)       # far fewer than a block: every deep round runs in slabs
        for bits in (32, 64):
            text = O.rand_dna(60011, 7)
            SA, ISA, LCP, rounds = same(mg, text, bits)
            ref = O.construct(text, bits=bits)
            assert np.array_equal(SA, ref["SA"]) and np.array_equal(ISA, ref["ISA"]) and np.array_equal(LCP, ref["LCP"])
            assert mg.memory()[1]
        # tandem repeat: buckets of n / 256 suffixes that cross rank boundaries, ~10 rounds in which every suffix is unresolved
        text = inputs.tandem(40000, 256, O.rand_dna(256, 3))
        for bits in (32, 64):
            SA, ISA, LCP, rounds = same(mg, text, bits)
            ref = O.construct(text, bits=bits)
            assert np.array_equal(SA, ref["SA"]) and np.array_equal(ISA, ref["ISA"]) and np.array_equal(LCP, ref["LCP"])
            peak, reduced, slab_rounds = mg.memory()
            assert reduced and slab_rounds >= 5
            # h doubles every round as in the one-step log; the counters of a slab round may run ahead of it
            assert [r[0] for r in rounds] == [t[0] for t in ref["trace"]][:len(rounds)]
        # forced bucket refinement from k = 3 (huge first buckets), without LCP
        text = O.rand_dna(30011, 23)
        SA, ISA, LCP, _ = same(mg, text, 64, k=3, lcp=False)
        assert LCP is None and np.array_equal(SA, O.naive_sa(text, 64))
        assert np.array_equal(ISA[SA.astype(np.int64)], np.arange(text.size, dtype=np.uint64))
        # low-entropy text: range minima over many buckets
        rng = np.random.RandomState(5)
        p = 0.5 ** np.arange(1, 21); p /= p.sum()
        text = (97 + rng.choice(20, size=200003, p=p)).astype(np.uint8)
        SA, ISA, LCP, rounds = same(mg, text, 32)
        ref = O.construct(text, bits=32)
        assert np.array_equal(SA, ref["SA"]) and np.array_equal(ISA, ref["ISA"]) and np.array_equal(LCP, ref["LCP"])
    finally:
        mg.close()


def test_multi_reduced_memory_bucket_longer_than_a_block():
    # A bucket of unresolved suffixes that covers a whole block (one symbol; a homopolymer run across three ranks) cannot be
    # cut into slabs whose parts on all its ranks meet in one step: such a round runs unsliced (suffix_array.hpp:1163-1212
    # sorts such buckets on sub-communicators), the other rounds stay in slabs.
    mg = multi(2)
    try:
        mg.configure(layout=mg.LAYOUT_REDUCED, slab=500)
        text = np.full(5003, 65, np.uint8)
        SA, ISA, LCP, _ = same(mg, text, 32)
        ref = O.construct(text, bits=32)
        assert np.array_equal(SA, ref["SA"]) and np.array_equal(ISA, ref["ISA"]) and np.array_equal(LCP, ref["LCP"])
        assert mg.memory()[1]
    finally:
        mg.close()
    mg = multi(3)
    try:
        mg.configure(layout=mg.LAYOUT_REDUCED, slab=300)
        rng = np.random.RandomState(4)
        text = np.frombuffer(b"ACGT", np.uint8)[rng.randint(0, 4, 9000)].copy()
        text[2500:6800] = 65                                   # the run covers rank 1's block [3000, 6000) and reaches into both neighbours
        for bits in (32, 64):
            SA, ISA, LCP, _ = same(mg, text, bits)
            ref = O.construct(text.tobytes(), bits=bits)
            assert np.array_equal(SA, ref["SA"]) and np.array_equal(ISA, ref["ISA"]) and np.array_equal(LCP, ref["LCP"])
            assert mg.memory()[1] and mg.memory()[2] > 0       # reduced layout, some rounds in slabs
    finally:
        mg.close()


def test_multi_reduced_memory_twin_of_config_c5_and_its_footprint():
    # configs[4] / 256 again (2^27 characters, period-1024 tandem, 8 ranks, uint64) in the reduced-memory layout, the
    # refinement rounds in slabs of 2^20 unresolved suffixes per rank (16 steps a round); verified by the distributed checker
    # and against the normal layout's result.  The footprint: the normal layout against the reduced one, in words per character.
    import ctypes as C
    P, n, bits = 8, 1 << 27, 64
    mg = multi(P)
    try:
        lib = mg._lib
        sizes = [n // P] * P
        slack = sizes[0] // 8 + 256
        ctxs = [mg.rank_ctx(i) for i in range(P)]
        def alloc(ctx, nbytes):
            p = C.c_void_p()
            assert lib.psacx_dev_alloc(ctx, C.byref(p), nbytes) == 0
            return p.value
        d_text = [alloc(c, m) for c, m in zip(ctxs, sizes)]
        for i, (c, m) in enumerate(zip(ctxs, sizes)):
            assert lib.psacx_synth_text_dev(c, C.c_void_p(d_text[i]), m, i * sizes[0], 2, 3, 1024) == 0
        outs = [[alloc(c, (m + slack) * 8) for c, m in zip(ctxs, sizes)] for _ in range(3)]
        res = {}
        for layout in (mg.LAYOUT_NORMAL, mg.LAYOUT_REDUCED):
            mg.configure(layout=layout, slab=1 << 20, output_slack=slack)
            st = mg.construct_device(d_text, sizes, outs[0], outs[1], outs[2], bits)[0]
            peak, reduced, slab_rounds = mg.memory()             # (before the checker allocates its own arrays)
            assert mg.check_device(d_text, sizes, outs[0], outs[1], outs[2], bits) == [0, 0, 0, 0]
            host = []
            for a in outs:
                parts = []
                for i, (c, m) in enumerate(zip(ctxs, sizes)):
                    h = np.empty(m, np.uint64)
                    assert lib.psacx_copy_d2h(c, h.ctypes.data_as(C.c_void_p), C.c_void_p(a[i]), m * 8) == 0
                    parts.append(h)
                host.append(np.concatenate(parts))
            res[layout] = (host, max(peak) / (sizes[0] * 8.0), reduced, slab_rounds, st.n_rounds)
        (a, wa, ra, sa_, na), (b, wb, rb, sb, nb) = res[mg.LAYOUT_NORMAL], res[mg.LAYOUT_REDUCED]
        assert all(np.array_equal(x, y) for x, y in zip(a, b))
        assert not ra and rb and sa_ == 0 and sb >= 10 and na >= 20 and nb >= 20
        print("words per character beside the results: normal %.2f, reduced %.2f" % (wa, wb))
        assert wb <= 5.0 and wb < wa
    finally:
        mg.close()


def test_multi_first_round_memory_in_the_reduced_layout():
    # psacx_multi_get_memory after a construction of random DNA in the reduced-memory layout (8 ranks x 2^28 characters, uint64, the
    # ranks sharing device 0): the one-word first round keeps the partitioned block, the receive array and the suffixes in the rank's three
    # result arrays and re-balances in place, so the engine's own allocations peak below 3 words per character (round 3: 4.75 with more
    # than one rank; psac plans 6 for its sort, idxsort.hpp:41-45).  With the result arrays (3 x 1.125) and the text that is the 6.5 words
    # a block of 2^32 characters is allowed on a 288 GiB part.  Verified by the distributed checker.  (Blocks of 2^28 characters: the
    # allocations that do not grow with the block -- wire pieces, sort tables, about 0.6 GB a rank -- are 1.2 words of a 2^26 block.)
    import ctypes as C
    P, m, bits = 8, 1 << 28, 64
    mg = multi(P)
    try:
        lib = mg._lib
        sizes = [m] * P
        slack = m // 8 + 256
        ctxs = [mg.rank_ctx(i) for i in range(P)]
        def alloc(ctx, nbytes):
            p = C.c_void_p()
            assert lib.psacx_dev_alloc(ctx, C.byref(p), nbytes) == 0
            return p.value
        d_text = [alloc(c, m) for c in ctxs]
        for i, c in enumerate(ctxs):
            assert lib.psacx_synth_text_dev(c, C.c_void_p(d_text[i]), m, i * m, 0, 1, 1024) == 0
        outs = [[alloc(c, (m + slack) * 8) for c in ctxs] for _ in range(3)]
        mg.configure(layout=mg.LAYOUT_REDUCED, output_slack=slack)
        mg.construct_device(d_text, sizes, outs[0], outs[1], outs[2], bits)
        peak, reduced, _ = mg.memory()
        form = mg.last_form()
        assert reduced and form["one_word"] and form["slice_inversion"]
        words = max(peak) / (m * 8.0)
        print("engine allocations at their peak: %.2f words per character" % words)
        assert words <= 3.0, words
        assert words + 3.0 * (m + slack) / m + 2.0 / 8 <= 6.5
        assert mg.check_device(d_text, sizes, outs[0], outs[1], outs[2], bits) == [0, 0, 0, 0]
        for c, ps in zip(ctxs, zip(d_text, *outs)):
            for p_ in ps:
                lib.psacx_dev_free(c, C.c_void_p(p_))
    finally:
        mg.close()


@pytest.mark.parametrize("P,lg", [(8, 28), (1, 30)])
def test_multi_repetitive_text_memory_in_the_reduced_layout(P, lg):
    # BASELINE.json configs[4] at reduced scale: a period-1024 tandem repeat (every suffix ties on the sorted prefix of the first round and
    # stays unresolved for ~27 rounds) in the reduced-memory layout, 8 ranks x 2^28 characters and one rank x 2^30, uint64.  The text keeps
    # the one-word first round (its ties are ordered slab by slab), the slice inversion of one rank runs in steps, a refinement step
    # releases what it no longer reads before the range minima: the engine's own allocations stay below 3 words per character (round 4:
    # the three-word fallback took 8.25 and a block of 2^32 characters did not fit the device; psac plans 6 words for the sort alone,
    # idxsort.hpp:41-45, and runs the same loop for any text, suffix_array.hpp:381-450).  Verified by the distributed checker.
    import ctypes as C
    m, bits = 1 << lg, 64
    mg = multi(P)
    try:
        lib = mg._lib
        sizes = [m] * P
        slack = m // 8 + 256
        ctxs = [mg.rank_ctx(i) for i in range(P)]
        def alloc(ctx, nbytes):
            p = C.c_void_p()
            assert lib.psacx_dev_alloc(ctx, C.byref(p), nbytes) == 0
            return p.value
        d_text = [alloc(c, m) for c in ctxs]
        for i, c in enumerate(ctxs):
            assert lib.psacx_synth_text_dev(c, C.c_void_p(d_text[i]), m, i * m, 2, 3, 1024) == 0
        outs = [[alloc(c, (m + slack) * 8) for c in ctxs] for _ in range(3)]
        mg.configure(layout=mg.LAYOUT_REDUCED, output_slack=slack)
        st = mg.construct_device(d_text, sizes, outs[0], outs[1], outs[2], bits)[0]
        peak, reduced, slab_rounds = mg.memory()
        form = mg.last_form()
        assert reduced and form["one_word"] and form["slice_inversion"] and form["tie_slabs"] >= 8 and slab_rounds >= 20 and st.n_rounds >= 24
        words = max(peak) / (m * 8.0)
        print("engine allocations at their peak: %.2f words per character" % words)
        assert words <= 3.0, words
        assert words + 3.0 * (m + slack) / m + 1.0 / 8 <= 6.5
        assert mg.check_device(d_text, sizes, outs[0], outs[1], outs[2], bits) == [0, 0, 0, 0]
        for c, ps in zip(ctxs, zip(d_text, *outs)):
            for p_ in ps:
                lib.psacx_dev_free(c, C.c_void_p(p_))
    finally:
        mg.close()


def test_multi_rccl_wire_forced_at_world_size_one(monkeypatch):
    # PSACX_MULTI_FORCE_WIRE=1: the shortcuts for data a rank addresses to itself and for scalars already on this host are
    # off, so ncclAllGather and ncclSend / ncclRecv (to self) are really issued at world size 1 -- the calls the driver's
    # 8-GPU run makes -- and the result is still the oracle's.  Both ways to build a communicator.
    import psac_amd
    monkeypatch.setenv("PSACX_MULTI_FORCE_WIRE", "1")
    text = O.rand_dna(150001, 11)
    ref = O.construct(text, bits=64)
    for how in ("rank", "all"):
        mg = psac_amd.MultiContext.for_rank(0, 1, 0, psac_amd.unique_id()) if how == "rank" else psac_amd.MultiContext([0])
        try:
            assert mg.transport == "rccl" and mg.uses_rccl
            SA, ISA, LCP, rounds = mg.construct(text, index_bits=64)
            assert np.array_equal(SA, ref["SA"]) and np.array_equal(ISA, ref["ISA"]) and np.array_equal(LCP, ref["LCP"])
            wire = mg.wire()
            assert wire["allgathers"] > 0 and wire["sends"] > 0 and wire["recvs"] == wire["sends"], wire
            assert wire["exchange_ms"][0] > 0
            # repetitive text: refinement rounds with B2 fetches, ISA updates and range minima through the wire
            t2 = inputs.tandem(30000, 256, O.rand_dna(256, 3))
            SA, ISA, LCP, rounds = mg.construct(t2, index_bits=32)
            r2 = O.construct(t2, bits=32)
            assert np.array_equal(SA, r2["SA"]) and np.array_equal(LCP, r2["LCP"])
            assert rounds == [(h, b, e) for h, b, e, _ in r2["trace"]]
        finally:
            mg.close()


def test_multi_rccl_wire_in_pieces(monkeypatch):
    # Messages travel in pieces of at most PSACX_MULTI_WIRE_PIECE bytes (default 2^28): a single ncclSend / ncclRecv of 2^31
    # bytes arrived damaged on this stack (profiles/r04k).  (a) pieces of 4 KiB on a small text against the oracle: every
    # message of the two-word shuffle, the tie windows and the slice inversion is cut many times; (b) 2^28 characters with
    # 64-bit indices on one rank with the wire forced: its messages to itself are 2^29 .. 2^31 bytes long (the checker's
    # fetches send a whole 2^31-byte array), verified by the distributed checker.
    import ctypes as C
    import psac_amd
    monkeypatch.setenv("PSACX_MULTI_FORCE_WIRE", "1")
    monkeypatch.setenv("PSACX_MULTI_TWO_WORD", "1")
    monkeypatch.setenv("PSACX_MULTI_WIRE_PIECE", "4096")
    text = O.rand_dna(300007, 5)
    ref = O.construct(text, bits=64)
    mg = psac_amd.MultiContext([0])
    try:
        SA, ISA, LCP, rounds = mg.construct(text, index_bits=64)
        assert mg.last_form()["two_word"]
        assert np.array_equal(SA, ref["SA"]) and np.array_equal(ISA, ref["ISA"]) and np.array_equal(LCP, ref["LCP"])
        assert mg.wire()["sends"] > 1000
    finally:
        mg.close()
    monkeypatch.delenv("PSACX_MULTI_WIRE_PIECE")
    monkeypatch.delenv("PSACX_MULTI_TWO_WORD")
    mg = psac_amd.MultiContext([0])
    try:
        lib, ctx, m = mg._lib, mg.rank_ctx(0), 1 << 28
        def alloc(nbytes):
            p = C.c_void_p()
            assert lib.psacx_dev_alloc(ctx, C.byref(p), nbytes) == 0
            return p.value
        d_text = [alloc(m)]
        assert lib.psacx_synth_text_dev(ctx, C.c_void_p(d_text[0]), m, 0, 0, 1, 1024) == 0
        outs = [[alloc(m * 8)] for _ in range(3)]
        mg.construct_device(d_text, [m], outs[0], outs[1], outs[2], 64)
        assert mg.transport == "rccl" and mg.last_form()["two_word"]
        assert mg.check_device(d_text, [m], outs[0], outs[1], outs[2], 64) == [0, 0, 0, 0]
        for p in d_text + [o[0] for o in outs]:
            lib.psacx_dev_free(ctx, C.c_void_p(p))
    finally:
        mg.close()


def _run_rank_processes(tmp_path, P, kind, n, seed, bits, extras=(), env_extra=None):
    """P processes, one rank each, all on device 0, exchanging through shared memory (PSACX_MULTI_TRANSPORT=shm)."""
    import json
    import os
    import subprocess
    import sys
    here = os.path.dirname(os.path.abspath(__file__))
    uid = os.urandom(128).hex()
    env = dict(os.environ, PSACX_MULTI_TRANSPORT="shm")
    env.update(env_extra or {})
    procs = [subprocess.Popen([sys.executable, os.path.join(here, "multi_rank_proc.py"), str(r), str(P), "0", uid, kind, str(n), str(seed), str(bits),
                               str(tmp_path)] + list(extras), env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True) for r in range(P)]
    outs = []
    for p in procs:
        try:
            o, e = p.communicate(timeout=600)
        except subprocess.TimeoutExpired:
            for q in procs:
                q.kill()
            raise
        outs.append((p.returncode, o, e))
    for rc, o, e in outs:
        assert rc == 0, e[-3000:]
    cat = lambda what: np.concatenate([np.load(os.path.join(str(tmp_path), "r%d_%s.npy" % (r, what))) for r in range(P)])
    info = [json.load(open(os.path.join(str(tmp_path), "r%d.json" % r))) for r in range(P)]
    return cat, info


@pytest.mark.parametrize("P,bits,kind,n", [(2, 32, "dna", 200003), (3, 64, "tandem", 90001), (2, 64, "single", 4099)])
def test_multi_process_per_rank_sharing_one_gpu(tmp_path, P, bits, kind, n):
    # psacx_multi_create_rank with L = 1 < P: one process per rank as under torchrun / mpirun, here with the host-staged
    # transport because RCCL refuses two ranks on one device.  The count all-gathers, rank(i) != i, the receive offsets by
    # source rank and the agreement on a status all run; a small box forces the streams through several rounds.
    import multi_rank_proc
    cat, info = _run_rank_processes(tmp_path, P, kind, n, 5, bits, extras=("check", "ansv"), env_extra={"PSACX_SHM_BOX": "65536"})
    text = multi_rank_proc.make_text(kind, n, 5)
    ref = O.construct(text, bits=bits)
    assert np.array_equal(cat("sa"), ref["SA"]) and np.array_equal(cat("isa"), ref["ISA"]) and np.array_equal(cat("lcp"), ref["LCP"])
    for r, inf in enumerate(info):
        assert inf["rank"] == r and inf["nranks"] == P and inf["nlocal"] == 1 and inf["transport"] == "shm"
        assert inf["check"] == [0, 0, 0, 0]
        assert [tuple(x) for x in inf["rounds"]] == [(h, b, e) for h, b, e, _ in ref["trace"]]
        assert inf["exchanges"] > 0 and inf["gathers"] > 0
    assert sum(inf["bytes_sent"] for inf in info) > 0
    # the distributed ANSV over the LCP blocks of the p processes (psac -t on p ranks, ansv.hpp:2042-2051)
    lcp = ref["LCP"].astype(np.uint64)
    assert np.array_equal(cat("left"), O.ansv(lcp, True, 2, n)) and np.array_equal(cat("right"), O.ansv(lcp, False, 0, n))


@pytest.mark.parametrize("P", [2, 5, 8])
def test_multi_blocks_shorter_than_the_kmer_window(P):
    # kmer.hpp:33-39 shrinks k to the smallest block; the 2k-character window of a position then reaches over several right
    # neighbours.  (Round 2 refused these inputs with PSACX_EINVAL.)
    mg = multi(P)
    try:
        for text, bits in ((O.as_text("mississippi"), 64), (O.rand_dna(3 * P + 1, 3), 32), (O.rand_dna(97, 5), 64), (O.as_text("a" * (P + 1)), 32),
                           (inputs.cyclic(61, "abc"), 64)):
            if text.size < P:
                continue
            SA, ISA, LCP, _ = same(mg, text, bits)
            ref = O.construct(text, bits=bits)
            assert np.array_equal(SA, ref["SA"]) and np.array_equal(ISA, ref["ISA"]) and np.array_equal(LCP, ref["LCP"]), (P, bits, text.size)
    finally:
        mg.close()


@pytest.mark.parametrize("P", [1, 2, 3, 7])
def test_multi_first_round_two_word_form(P, monkeypatch):
    # sort_first_two_word (multi.hpp): the shuffle and the local sort move (word 1, suffix) only, on the leading bits; the
    # suffixes that tie fetch their full window from the ranks that own their text.  Forced below its size threshold;
    # mode 2 also sends repetitive texts through it (long tie groups -> radix sort of the compacted ties).
    cases = [(O.rand_dna(70001, 7), 64), (O.rand_dna(70001, 7), 32), (inputs.ascii128(50000, 3), 64), (inputs.tandem(30000, 256, O.rand_dna(256, 3)), 64),
             (np.full(5003, 65, np.uint8), 64), (inputs.cyclic(20011, "abc"), 32), (O.as_text("mississippi" * 40), 64)]
    # shuffle by key ranges with one sort per range under the exchanges, also with other numbers of ranges
    monkeypatch.setenv("PSACX_MULTI_ONE_WORD", "0")
    for mode, env in (("1", {}), ("2", {}), ("2", {"PSACX_MULTI_PIECES": "7"}), ("1", {"PSACX_MULTI_PIECES": "1"})):
        monkeypatch.setenv("PSACX_MULTI_TWO_WORD", mode)
        for k_ in ("PSACX_MULTI_PIECES",):
            monkeypatch.delenv(k_, raising=False)
        for k_, v_ in env.items():
            monkeypatch.setenv(k_, v_)
        mg = multi(P)
        try:
            used = 0
            for text, bits in cases:
                SA, ISA, LCP, rounds = same(mg, text, bits)
                ref = O.construct(text, bits=bits)
                assert np.array_equal(SA, ref["SA"]) and np.array_equal(ISA, ref["ISA"]) and np.array_equal(LCP, ref["LCP"]), (P, mode, env, bits, text.size)
                assert rounds == [(h, b, e) for h, b, e, _ in ref["trace"]]
                used += mg.last_form()["two_word"]
            assert used >= (len(cases) - 1 if mode == "2" else 2), used      # (DNA on 32-bit words: word 1 is shorter than the leading bits)
        finally:
            mg.close()


@pytest.mark.parametrize("P", [1, 2, 3, 7])
def test_multi_first_round_one_word_form(P, monkeypatch):
    # sort_first_one_word (multi.hpp): the 256 buckets of the top digit of the prefix are dealt whole to the ranks from exact counts,
    # the sender writes one-word records (rest of the prefix | suffix) straight from the text, the buckets travel in groups and are
    # sorted as they land (radix_scatter1w_kernel), the short suffixes are made on the host.  Forced below its size threshold
    # (PSACX_MULTI_ONE_WORD=1); PSACX_MULTI_TWO_WORD=2 also sends repetitive and badly balanced texts through it (long tie groups,
    # buckets of one rank only, empty ranks).
    cases = [(O.rand_dna(70001, 7), 64), (inputs.ascii128(50000, 3), 64), (inputs.tandem(30000, 256, O.rand_dna(256, 3)), 64),
             (np.full(5003, 65, np.uint8), 64), (inputs.cyclic(20011, "abc"), 64), (O.as_text("mississippi" * 40), 64),
             (O.rand_dna(70001, 7), 32)]
    monkeypatch.setenv("PSACX_MULTI_ONE_WORD", "1")
    for mode, env in (("1", {}), ("2", {}), ("2", {"PSACX_MULTI_PIECES": "7"}), ("1", {"PSACX_MULTI_PIECES": "1"}), ("1", {"PSACX_MULTI_FORCE_WIRE": "1"})):
        if env.get("PSACX_MULTI_FORCE_WIRE") and P != 1:
            continue
        monkeypatch.setenv("PSACX_MULTI_TWO_WORD", mode)
        for k_ in ("PSACX_MULTI_FORCE_WIRE", "PSACX_MULTI_PIECES"):
            monkeypatch.delenv(k_, raising=False)
        for k_, v_ in env.items():
            monkeypatch.setenv(k_, v_)
        mg = multi(P)
        try:
            used = 0
            for text, bits in cases:
                SA, ISA, LCP, rounds = same(mg, text, bits)
                ref = O.construct(text, bits=bits)
                assert np.array_equal(SA, ref["SA"]) and np.array_equal(ISA, ref["ISA"]) and np.array_equal(LCP, ref["LCP"]), (P, mode, env, bits, text.size)
                assert rounds == [(h, b, e) for h, b, e, _ in ref["trace"]]
                used += mg.last_form()["one_word"]
            assert used >= (len(cases) - 2 if mode == "2" else 2), (used, P, mode)      # (32-bit words never take the form, nor do blocks shorter than 128 characters)
        finally:
            mg.close()


@pytest.mark.parametrize("P", [1, 2, 3, 7])
def test_multi_first_round_ties_in_slabs(P, monkeypatch):
    # Reduced-memory layout: a repetitive text stays in one-word records (sort_first_one_word does not turn it away) and its ties --
    # every suffix of a tandem repeat -- are ordered slab by slab (first_sort_ties): slabs end where a group of equal prefixes ends, a
    # group longer than a slab (one symbol, a period of three) is taken whole, ranks without ties left run empty slabs along.
    # BASELINE.json configs[4] at reduced scale; idxsort.hpp:41-45 plans the whole second record set instead.
    cases = [(inputs.tandem(30000, 256, O.rand_dna(256, 3)), 64), (O.rand_dna(70001, 7), 64), (np.full(5003, 65, np.uint8), 64),
             (inputs.cyclic(20011, "abc"), 64), (O.as_text("mississippi" * 40), 64), (inputs.tandem(50021, 1024, O.rand_dna(1024, 5)), 64)]
    monkeypatch.setenv("PSACX_MULTI_ONE_WORD", "1")
    monkeypatch.setenv("PSACX_MULTI_TWO_WORD", "1")
    for slab, env in ((700, {}), (4000, {"PSACX_MULTI_PIECES": "3"}), (700, {"PSACX_MULTI_FORCE_WIRE": "1"})):
        if env.get("PSACX_MULTI_FORCE_WIRE") and P != 1:
            continue
        for k_ in ("PSACX_MULTI_FORCE_WIRE", "PSACX_MULTI_PIECES"):
            monkeypatch.delenv(k_, raising=False)
        for k_, v_ in env.items():
            monkeypatch.setenv(k_, v_)
        mg = multi(P)
        try:
            mg.configure(layout=mg.LAYOUT_REDUCED, slab=slab)
            slabs = []
            for text, bits in cases:
                SA, ISA, LCP, rounds = same(mg, text, bits)
                ref = O.construct(text, bits=bits)
                assert np.array_equal(SA, ref["SA"]) and np.array_equal(ISA, ref["ISA"]) and np.array_equal(LCP, ref["LCP"]), (P, slab, env, text.size)
                form = mg.last_form()
                assert form["reduced_memory"]
                slabs.append((form["one_word"], form["tie_slabs"]))
            # the tandem repeats took the one-word form although every suffix ties, and their ties needed several slabs
            assert slabs[0][0] and slabs[0][1] >= 1 and slabs[5][0] and slabs[5][1] >= 1, slabs
            assert slabs[1][1] == 0, slabs              # random text: a handful of ties, one slab
        finally:
            mg.close()


@pytest.mark.parametrize("P", [1, 2, 3, 7])
def test_multi_string_sets(P):
    # construct_ss on p ranks (suffix_array.hpp:267-363; psacx_multi_construct_gsa_*): the reference's expected arrays
    # (test/test_gsa.cpp:35-105), random sets against the oracle's restatement, deep ties (equal strings, prefixes of each
    # other: text order among equal suffixes), strings longer than a block and shorter than the k-mer window
    rng = np.random.RandomState(11)
    sets = []
    for sigma, m, lo, hi in ((4, 300, 1, 400), (2, 50, 1, 30), (1, 40, 1, 100), (26, 2000, 5, 60), (4, 1, 5000, 5001), (4, 3000, 1, 3), (90, 200, 100, 2000)):
        sets.append([bytes(rng.randint(65, 65 + sigma, size=int(rng.randint(lo, hi))).astype(np.uint8)) for _ in range(m)])
    base = bytes(inputs.dna(300, 4))
    sets.append([base[:int(x)] for x in rng.randint(1, 300, size=500)])
    sets.append([base] * 200)
    mg = multi(P)
    try:
        from test_oracle_golden import GSA_REPEATS, repeat_inc_gsa, repeat_inc_glcp, repeat_inc_seq
        for bits in (64, 32):
            SA, ISA, LCP, _, _ = mg.construct_ss(["abab", "baba"], index_bits=bits)            # test/test_gsa.cpp:73-105 (SimpleTiny)
            assert SA.tolist() == [7, 2, 5, 0, 3, 6, 1, 4] and LCP.tolist() == [0, 1, 2, 3, 0, 1, 2, 3]
            for seq, reps in GSA_REPEATS:                                                      # test/test_gsa.cpp:107-179 (IncRepeats*)
                SA, ISA, LCP, _, _ = mg.construct_ss(repeat_inc_seq(seq, reps), index_bits=bits)
                assert SA.tolist() == repeat_inc_gsa(len(seq), reps) and LCP.tolist() == repeat_inc_glcp(len(seq), reps), (seq, reps, bits)
                assert np.array_equal(ISA[SA.astype(np.int64)], np.arange(SA.size, dtype=SA.dtype))
        for strings in sets:
            for bits, k in ((32, 0), (64, 0), (32, 3)):
                SA, ISA, LCP, rounds, _ = mg.construct_ss(strings, index_bits=bits, k=k)
                ref = O.construct_ss(strings, bits=bits, k=k)
                assert np.array_equal(SA, ref["SA"]) and np.array_equal(ISA, ref["ISA"]) and np.array_equal(LCP, ref["LCP"]), (P, bits, k, len(strings))
            SA, ISA, LCP, _, _ = mg.construct_ss(strings, index_bits=32, lcp=False)
            assert LCP is None and np.array_equal(SA, ref["SA"].astype(np.uint32))
        with pytest.raises(Exception):
            mg._lib.psacx_multi_construct_gsa_u64.restype = int
            t = np.frombuffer(b"abcabc", np.uint8); off = np.array([0, 4, 3, 6], np.uint64)
            SA = np.empty(6, np.uint64)
            p_ = lambda a: a.ctypes.data_as(__import__("ctypes").c_void_p)
            mg.check(mg._lib.psacx_multi_construct_gsa_u64(mg.handle, p_(t), 6, p_(off), 3, 0, 0, p_(SA), p_(SA.copy()), None))
    finally:
        mg.close()


@pytest.mark.parametrize("P", [1, 2, 3, 7])
def test_multi_suffix_tree_node_table(P):
    # psacx_multi_suffix_tree_dev_*: construct_suffix_tree on p ranks (suffix_tree.hpp:413-499) -- the rows of the node table
    # block-distributed like LCP, against the reference's mississippi table (test/test_suffixtree.cpp:68-83) and the oracle's
    # one-rank restatement for the shapes of test/test_suffixtree.cpp:89-162 (random DNA, (abc)^n) and a larger alphabet
    import ctypes as C
    import json
    import os
    kat = json.load(open(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "reference_kat.json")))["mississippi"]
    mg = multi(P)
    lib = mg._lib
    try:
        cases = [(O.as_text(kat["text"]), 64), (O.rand_dna(116, 13), 64), (O.rand_dna(1000, 13), 32), (O.rand_dna(23713, 13), 64),
                 (inputs.cyclic(3000, "abc"), 32), (inputs.ascii128(5000, 2), 64), (inputs.tandem(4000, 64, O.rand_dna(64, 3)), 32)]
        for text, bits in cases:
            n = text.size
            if n < P:
                continue
            w = bits // 8
            udt = np.uint32 if bits == 32 else np.uint64
            ref = O.construct(text, bits=bits)
            want = O.suffix_tree(text, ref["SA"], ref["LCP"])
            sizes = [n // P + (1 if r < n % P else 0) for r in range(P)]
            offs = np.concatenate([[0], np.cumsum(sizes)]).astype(np.int64)
            held = []

            def alloc(ctx, nbytes):
                p = C.c_void_p()
                assert lib.psacx_dev_alloc(ctx, C.byref(p), max(int(nbytes), 1)) == 0
                held.append((ctx, p))
                return p.value
            d_t, d_sa, d_lcp = [], [], []
            for r in range(P):
                ctx = mg.rank_ctx(r)
                d_t.append(alloc(ctx, sizes[r])); d_sa.append(alloc(ctx, sizes[r] * w)); d_lcp.append(alloc(ctx, sizes[r] * w))
                for dst, arr in ((d_t[r], text), (d_sa[r], ref["SA"].astype(udt)), (d_lcp[r], ref["LCP"].astype(udt))):
                    blk = np.ascontiguousarray(arr[offs[r]:offs[r + 1]])
                    assert lib.psacx_copy_h2d(ctx, C.c_void_p(dst), blk.ctypes.data_as(C.c_void_p), blk.nbytes) == 0
            sigma = mg.suffix_tree_device(d_t, sizes, None, None, None, bits)
            assert sigma == want.shape[1] - 1
            d_nodes = [alloc(mg.rank_ctx(r), sizes[r] * (sigma + 1) * 8) for r in range(P)]
            assert mg.suffix_tree_device(d_t, sizes, d_sa, d_lcp, d_nodes, bits) == sigma
            got = np.empty((n, sigma + 1), np.uint64)
            for r in range(P):
                blk = got[offs[r]:offs[r + 1]]
                assert lib.psacx_copy_d2h(mg.rank_ctx(r), blk.ctypes.data_as(C.c_void_p), C.c_void_p(d_nodes[r]), blk.nbytes) == 0
            assert np.array_equal(got, want), (P, bits, n)
            if text.size == len(kat["text"]):
                assert got.reshape(-1).tolist() == kat["suffix_tree_nodes"]
            for ctx, p in held:
                lib.psacx_dev_free(ctx, p)
    finally:
        mg.close()


@pytest.mark.parametrize("P,wb,s1,step", [(1, 6, 9, 0), (2, 5, 2, 1), (3, 14, 9, 0), (7, 4, 1, 1), (4, 6, 3, 2)])
def test_multi_isa_by_destination_slices(P, wb, s1, step, monkeypatch):
    # SA -> ISA of the first round slice by slice (slice_inv.hpp): first level by (owner, slice) on the senders, the slices
    # travel in steps (double-buffered), further reservation levels + the LDS window scatter on the owners.  Small windows
    # and few slice bits force the deeper levels on test-sized inputs.
    monkeypatch.setenv("PSACX_SLICE_SHAPE", "%d,%d,%d" % (wb, s1, step))
    # (the pairs travel and are partitioned as packed 64-bit entries; PSACX_SLICE_WIDE=1: the 64-bit rank form of texts beyond
    #  2^32 characters, whose ranks travel as 32 bits relative to the end of the sender's block and are widened by the first
    #  owner-side kernel)
    for two in (False, "wide"):
        if two == "wide":
            monkeypatch.setenv("PSACX_SLICE_WIDE", "1")
        mg = multi(P)
        try:
            for text, bits in ((O.rand_dna(300007, 7), 64), (O.rand_dna(131072 * P, 9), 32), (inputs.tandem(90001, 256, O.rand_dna(256, 3)), 32),
                               (inputs.ascii128(70000, 3), 64), (O.as_text("mississippi"), 64), (np.full(9001, 65, np.uint8), 32)):
                SA, ISA, LCP, rounds = same(mg, text, bits)
                ref = O.construct(text, bits=bits)
                assert np.array_equal(SA, ref["SA"]) and np.array_equal(ISA, ref["ISA"]) and np.array_equal(LCP, ref["LCP"]), (P, bits, text.size, two)
                assert mg.last_form()["slice_inversion"]
        finally:
            mg.close()
