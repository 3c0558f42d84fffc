"""GPU parity: the HIP engine (through the C-ABI of libpsacx.so) against the CPU
oracle and the reference's golden vectors, bit-exact.  Run with -m gpu on an MI355X."""
import json
import os
import sys

import numpy as np
import pytest

import inputs
import oracle_lib as O

pytestmark = pytest.mark.gpu

HERE = os.path.dirname(os.path.abspath(__file__))
KAT = json.load(open(os.path.join(HERE, "golden", "reference_kat.json")))


@pytest.fixture(scope="module")
def ctx():
    import psac_amd
    c = psac_amd.Context(0)
    yield c
    c.close()


def run(ctx, text, bits=32, lcp=True, fast=True, k=0):
    import psac_amd
    sa = psac_amd.SuffixArray(index_bits=bits, lcp=lcp, ctx=ctx)
    sa.construct(text, fast_resolval=fast, k=k)
    return sa


def same_as_oracle(ctx, text, bits=32, fast=True, k=0):
    got = run(ctx, text, bits=bits, lcp=True, fast=fast, k=k)
    ref = O.construct(text, bits=bits, fast=fast, k=k)
    assert got.k == ref["k"] and got.bits_per_char == ref["l"]
    assert np.array_equal(got.local_SA, ref["SA"])
    assert np.array_equal(got.local_B, ref["ISA"])
    assert np.array_equal(got.local_LCP, ref["LCP"])
    return got, ref


def test_mississippi_known_answer(ctx):
    m = KAT["mississippi"]
    for bits in (32, 64):
        sa = run(ctx, m["text"], bits=bits)
        assert sa.local_SA.tolist() == m["SA"]          # test/test_psac.cpp:105
        assert sa.local_B.tolist() == m["ISA"]
        assert sa.local_LCP.tolist() == m["LCP"]


def _make(row):
    from test_oracle_golden import make_input
    return make_input(row)


@pytest.mark.parametrize("row", KAT["checksums"], ids=[r["name"] for r in KAT["checksums"]])
def test_reference_checksums(ctx, row):
    text = _make(row)
    bits = 64 if row["name"] in ("rand_dna_66763_23", "ascii128_1M_42", "bytes127_1M_42") else 32
    sa = run(ctx, text, bits=bits)
    assert "%016x" % O.fnv(sa.local_SA) == row["sa"]
    assert "%016x" % O.fnv(sa.local_B) == row["isa"]
    assert "%016x" % O.fnv(sa.local_LCP) == row["lcp"]
    assert int(sa.local_LCP.max()) == row["max_lcp"]


@pytest.mark.parametrize("bits,fast,k", [(32, True, 0), (32, True, 3), (32, False, 2), (64, True, 0),
                                         (64, True, 3), (64, False, 0)])
def test_rand_all_variants(ctx, bits, fast, k):
    # test/test_psac.cpp:131-176 (RandAll, n = 130370, rand_dna(size, 7))
    text = O.rand_dna(130370, 7)
    got, ref = same_as_oracle(ctx, text, bits=bits, fast=fast, k=k)
    # iteration log parity: the (h, unfinished buckets, unfinished elements) sequence is a
    # property of the text; the oracle prints it from the doubling loop and the chasing loop
    o = [(h, b, e) for (h, b, e, _) in ref["trace"]]
    g = [(h, b, e) for (h, b, e, *_rest) in got.rounds]
    assert g == o


def test_lcp1(ctx):
    # test/test_psac.cpp:250-274
    text = O.rand_dna(66763, 23)
    same_as_oracle(ctx, text, bits=64)
    same_as_oracle(ctx, text, bits=64, k=3)


def test_repeats(ctx):
    # test/test_psac.cpp:178-224
    words = ["helloworld", "blahlablah", "ellow", "worldblah", "rld", "hello"]
    rng = np.random.default_rng(3)
    s = "".join(words[i] for i in rng.integers(0, len(words), 15000))
    for k in (0, 3):
        same_as_oracle(ctx, s, bits=64, k=k)
    same_as_oracle(ctx, s, bits=64, fast=False, k=2)


def test_small_and_degenerate(ctx):
    # test/test_psac.cpp:226-248 (n = 9) and the sizes around the k clamp (kmer.hpp:33-38)
    for n in (2, 3, 4, 5, 9, 10, 11, 17, 21, 22, 63, 64, 65, 255, 256, 257, 2047, 2048, 2049, 4097):
        same_as_oracle(ctx, O.rand_dna(n, 13), bits=32)
        same_as_oracle(ctx, O.rand_dna(n, 13), bits=64)
    sa = run(ctx, b"A", bits=32)
    assert sa.local_SA.tolist() == [0] and sa.local_B.tolist() == [0] and sa.local_LCP.tolist() == [0]
    for s in (b"A" * 1000, b"AB" * 700, b"\x00\x00\x01\x00\x00", bytes(range(256)) * 3,
              bytes(range(255, -1, -1)) * 5, b"abcabcabcabcabcabcabcabcabcabcabcabc"):
        same_as_oracle(ctx, s, bits=32)
        same_as_oracle(ctx, s, bits=64)


def test_ragged_tile_sizes(ctx):
    # sizes that leave partial radix / scan tiles
    for n in (4095, 4096, 4097, 8191, 12289, 100003):
        text = inputs.dna(n, 5)
        same_as_oracle(ctx, text, bits=32)
    same_as_oracle(ctx, inputs.ascii128(70001, 9), bits=64)
    same_as_oracle(ctx, inputs.bytes_mod127p1(50021, 2), bits=32)


def test_tandem_many_rounds(ctx):
    # 17 rounds, 1024 unfinished buckets until the last
    row = [r for r in KAT["checksums"] if r["name"] == "tandem_1M_1024_3"][0]
    text = _make(row)
    sa = run(ctx, text, bits=32)
    assert len(sa.rounds) == 17
    n = row["n"]
    for i, r in enumerate(sa.rounds):
        h, ub, ue = r[0], r[1], r[2]
        assert h == 10 << i
        if i < 16:
            assert ub == 1024 and ue == n - 2 * h + 1
        else:
            assert ub == 0 and ue == 0


def test_no_lcp_mode(ctx):
    text = O.rand_dna(50000, 3)
    sa = run(ctx, text, bits=32, lcp=False)
    ref = O.construct(text, bits=32, lcp=False)
    assert np.array_equal(sa.local_SA, ref["SA"]) and np.array_equal(sa.local_B, ref["ISA"])
    assert sa.local_LCP.size == 0


def test_medium_dna_properties(ctx):
    # 16 Mi characters: order property + Kasai on the host (check_suffix_array.hpp:56-88, lcp.hpp:46-77)
    n = 1 << 24
    text = inputs.dna(n, 1)
    sa = run(ctx, text, bits=32)
    assert O.check_sa(text, sa.local_SA, sa.local_B) == 0
    assert np.array_equal(O.kasai(text, sa.local_SA, sa.local_B), sa.local_LCP)


def test_isa_inversion_partition_path(ctx):
    # sizes >= 2^22 take the destination-partition path of the SA -> ISA inversion
    # (bulk_permute.hpp:14-73); ragged sizes leave partial buckets / windows
    for n, bits in (((1 << 22) + 12345, 32), ((1 << 22) + 4097, 64), ((1 << 23) - 1, 32)):
        text = inputs.dna(n, 4)
        sa = run(ctx, text, bits=bits)
        assert O.check_sa(text, sa.local_SA, sa.local_B) == 0
        assert np.array_equal(O.kasai(text, sa.local_SA, sa.local_B), sa.local_LCP)


def test_isa_inversion_forms(ctx, monkeypatch):
    # The SA -> ISA inversion by destination-partition levels + LDS window scatter: 64-bit words with the first level fused into
    # rebucket_first_kernel and packed (position | rank) pairs, 32-bit words with the fused two-array first level (normal layout) and
    # with radix levels (reduced-memory layout): property check + Kasai, and the two layouts give identical arrays.
    cases = (((1 << 22) + 4097, 64, 5), ((1 << 23) + 77, 32, 6), ((1 << 22) + 1, 32, 7))
    base = []
    for n, bits, seed in cases:
        text = inputs.dna(n, seed)
        sa = run(ctx, text, bits=bits)
        assert O.check_sa(text, sa.local_SA, sa.local_B) == 0
        assert np.array_equal(O.kasai(text, sa.local_SA, sa.local_B), sa.local_LCP)
        base.append((sa.local_SA.copy(), sa.local_B.copy(), sa.local_LCP.copy()))
    monkeypatch.setenv("PSACX_FORCE_DIET", "1")
    for (n, bits, seed), (SA, B, LCP) in zip(cases, base):
        sa = run(ctx, inputs.dna(n, seed), bits=bits)
        assert np.array_equal(sa.local_SA, SA) and np.array_equal(sa.local_B, B) and np.array_equal(sa.local_LCP, LCP)


def test_pair_sort_standalone(ctx):
    # idxsort.hpp:23-83: records (b1, b2, i) sorted by (b1, b2)
    import ctypes as C
    rng = np.random.default_rng(1)
    for bits, dt in ((32, np.uint32), (64, np.uint64)):
        for n in (1, 2, 1000, 4096, 4097, 300001):
            hi = 1 << 20
            b1 = rng.integers(0, 50, n).astype(dt)
            b2 = rng.integers(0, hi, n).astype(dt)
            d1 = ctx.alloc(b1.nbytes); d2 = ctx.alloc(b2.nbytes); di = ctx.alloc(b1.nbytes)
            ctx.h2d(d1, b1); ctx.h2d(d2, b2)
            fn = getattr(ctx._lib, "psacx_pair_sort_dev_u%d" % bits)
            ctx.check(fn(ctx.handle, C.c_void_p(d1), C.c_void_p(d2), C.c_void_p(di), n, 21))
            o1 = np.empty(n, dt); o2 = np.empty(n, dt); oi = np.empty(n, dt)
            ctx.d2h(o1, d1); ctx.d2h(o2, d2); ctx.d2h(oi, di)
            for p in (d1, d2, di):
                ctx.free(p)
            order = np.lexsort((b2, b1))          # stable, so ties keep index order
            assert np.array_equal(oi, order.astype(dt))
            assert np.array_equal(o1, b1[order]) and np.array_equal(o2, b2[order])


def test_errors(ctx):
    import psac_amd
    sa = psac_amd.SuffixArray(index_bits=32, lcp=True, ctx=ctx)
    with pytest.raises(ValueError):
        sa.construct(b"")
    import ctypes as C
    rc = ctx._lib.psacx_construct_u32(ctx.handle, None, 5, 0, 0, None, None, None)
    assert rc == -1


def test_ansv_all_type_combinations(ctx):
    # test/test_ansv.cpp:232-252, 270-282: every (left_type, right_type) on rand() % 100 inputs,
    # n in {13, 137, 1000, 26666}; plus an LCP array (suffix_tree.hpp:62 uses furthest_eq / nearest_sm)
    import psac_amd
    rng = np.random.default_rng(17)
    NO = 2**64 - 1
    cases = [rng.integers(0, 100, n).astype(np.uint32) for n in (1, 2, 13, 137, 1000, 26666)]
    cases.append(rng.integers(0, 3, 70000).astype(np.uint64))
    cases.append(np.zeros(5000, np.uint32))
    cases.append(np.arange(5000, dtype=np.uint32))
    cases.append(np.arange(5000, dtype=np.uint64)[::-1].copy())
    text = O.rand_dna(200000, 5)
    cases.append(O.construct(text, bits=32)["LCP"])
    for v in cases:
        for lt in (0, 1, 2):
            for rt in (0, 1, 2):
                if v.size > 30000 and (lt, rt) not in ((0, 0), (2, 0), (1, 2)):
                    continue
                left, right = psac_amd.ansv(v, lt, rt, nonsv=NO, ctx=ctx)
                assert np.array_equal(left, O.ansv(v, True, lt, NO)), (v.size, lt, rt)
                assert np.array_equal(right, O.ansv(v, False, rt, NO)), (v.size, lt, rt)


def test_cli_and_cpp_header(ctx, tmp_path):
    # psac CLI parity (src/psac.cpp:65-128): -f/-l/-c/-o, .sa64/.lcp64 as raw uint64, print64 listing
    # of README.md:88-100
    import subprocess
    root = os.path.dirname(HERE)
    psac = os.path.join(root, "psac_amd", "bin", "psac")
    p64 = os.path.join(root, "psac_amd", "bin", "print64")
    if not (os.path.exists(psac) and os.path.exists(p64)):
        pytest.skip("CLI not built")
    f = tmp_path / "miss.txt"
    f.write_bytes(b"mississippi")
    r = subprocess.run([psac, "-f", str(f), "-l", "-c", "-o", str(tmp_path / "out")], capture_output=True, text=True)
    assert r.returncode == 0, r.stderr
    assert "PSAC time:" in r.stderr and "[SUCCESS]" in r.stderr
    sa = np.fromfile(str(tmp_path / "out.sa64"), dtype=np.uint64)
    lcp = np.fromfile(str(tmp_path / "out.lcp64"), dtype=np.uint64)
    assert sa.tolist() == KAT["mississippi"]["SA"] and lcp.tolist() == KAT["mississippi"]["LCP"]
    listing = subprocess.run([p64, str(tmp_path / "out.sa64")], capture_output=True, text=True).stdout.split()
    assert [int(x) for x in listing] == KAT["mississippi"]["SA"]
    # random input through -r, checked by the CLI's own -c, then the -t path (ANSV over LCP)
    r = subprocess.run([psac, "-r", "300000", "-s", "3", "-l", "-c"], capture_output=True, text=True)
    assert r.returncode == 0 and "[SUCCESS]" in r.stderr, r.stderr
    r = subprocess.run([psac, "-r", "100000", "-t", "-c"], capture_output=True, text=True)
    assert r.returncode == 0 and "ST time:" in r.stderr and "ST edges:" in r.stderr, r.stderr
    # argument errors exit non-zero like TCLAP (src/psac.cpp:147-150)
    assert subprocess.run([psac], capture_output=True).returncode != 0
    assert subprocess.run([psac, "-f", str(f), "-r", "5"], capture_output=True).returncode != 0


def _dist_loopback_gpu(text, P, bits, k=0):
    import torch
    from dist_harness import dist as D
    from dist_harness.comm import LoopbackWorld
    from dist_harness.dist_ops import HipOps
    sizes = D.blk_sizes(text.size, P)
    offs = D.prefix(sizes)
    ops = [HipOps(bits, 0) for _ in range(P)]
    blocks = [torch.from_numpy(text[o:o + s].copy()).cuda() for o, s in zip(offs, sizes)]

    def fn(comm, op, blk):
        return (yield from D.construct(comm, op, blk, want_lcp=True, k_req=k))
    res = LoopbackWorld(P).run(fn, [(ops[r], blocks[r]) for r in range(P)])
    udt = np.uint32 if bits == 32 else np.uint64
    cat = lambda key: np.concatenate([r[key].cpu().numpy().view(udt) for r in res])
    out = cat("SA"), cat("ISA"), cat("LCP"), res[0]["rounds"]
    for o in ops:
        o.close()
    return out


@pytest.mark.parametrize("P", [1, 2, 3, 4])
def test_distributed_ops_on_one_gpu(ctx, P):
    # the block-distributed choreography with the HIP step ops, P virtual ranks sharing this GPU
    for bits in (32, 64):
        text = O.rand_dna(60011, 7)
        sa, isa, lcp, _ = _dist_loopback_gpu(text, P, bits)
        ref = O.construct(text, bits=bits)
        assert np.array_equal(sa, ref["SA"]) and np.array_equal(isa, ref["ISA"]) and np.array_equal(lcp, ref["LCP"])
    unit = O.rand_dna(256, 3)
    text = inputs.tandem(40000, 256, unit)
    sa, isa, lcp, rounds = _dist_loopback_gpu(text, P, 32)
    ref = O.construct(text, bits=32)
    assert np.array_equal(sa, ref["SA"]) and np.array_equal(isa, ref["ISA"]) and np.array_equal(lcp, ref["LCP"])
    text = O.rand_dna(30011, 23)
    sa, isa, lcp, _ = _dist_loopback_gpu(text, P, 64, k=3)
    assert np.array_equal(sa, O.naive_sa(text, 64))
    assert np.array_equal(O.kasai(text, sa, isa), lcp)


def test_distributed_shift_saturates(ctx):
    # psacx_op_add_scalar: SA + h in 64 bits, clamped to n (see tests/test_dist_cpu.py for the CPU twin)
    import torch
    from dist_harness.dist_ops import HipOps
    ops = HipOps(32, 0)
    n = 0xFFFFFF00
    sa = torch.from_numpy(np.array([5, 0x80000000, 0xFFFFFE00, 0xFFFFFEFF], np.uint32).view(np.int32)).cuda()
    q = ops.add_scalar(sa, 0x200, n)
    assert q.cpu().numpy().view(np.uint32).tolist() == [0x205, 0x80000200, n, n]
    ops.close()


def test_distributed_ops_larger(ctx):
    text = inputs.dna((1 << 22) + 1234, 9)
    sa, isa, lcp, _ = _dist_loopback_gpu(text, 3, 32)
    assert O.check_sa(text, sa, isa) == 0
    assert np.array_equal(O.kasai(text, sa, isa), lcp)


def test_low_entropy_text_many_range_minima(ctx):
    # 2^21 characters with geometric symbol frequencies: most suffixes stay unresolved after the first round, so
    # the refinement issues ~10^6 range minima per round and the running-minimum tables (levels 0 and up) are in
    # use -- on the single-GPU engine and in the distributed range_min op
    rng = np.random.RandomState(5)
    p = 0.5 ** np.arange(1, 21); p /= p.sum()
    text = (97 + rng.choice(20, size=(1 << 21) + 77, p=p)).astype(np.uint8)
    ref = O.construct(text, bits=32)
    got = run(ctx, text, bits=32)
    assert np.array_equal(got.local_SA, ref["SA"]) and np.array_equal(got.local_B, ref["ISA"]) and np.array_equal(got.local_LCP, ref["LCP"])
    assert [(h, b, e) for h, b, e, *_ in got.rounds] == [(h, b, e) for h, b, e, _ in ref["trace"]]
    wide = run(ctx, text, bits=64)
    assert np.array_equal(wide.local_LCP, ref["LCP"].astype(np.uint64))
    sa, isa, lcp, _ = _dist_loopback_gpu(text, 2, 32)
    assert np.array_equal(sa, ref["SA"]) and np.array_equal(isa, ref["ISA"]) and np.array_equal(lcp, ref["LCP"])


def _device_run_and_check(ctx, text, bits, corrupt=False):
    import psac_amd
    n = int(text.size); w = bits // 8
    d_text = ctx.alloc(n); ctx.h2d(d_text, text)
    d_sa, d_isa, d_lcp = ctx.alloc(n * w), ctx.alloc(n * w), ctx.alloc(n * w)
    sa = psac_amd.SuffixArray(index_bits=bits, lcp=True, ctx=ctx)
    sa.construct_device(d_text, n, d_sa, d_isa, d_lcp)
    if corrupt:
        bad = np.array([5], np.uint32 if bits == 32 else np.uint64)
        ctx.h2d(d_lcp + 1000 * w, bad + 77)
    err = psac_amd.check_device(ctx, d_text, n, d_sa, d_isa, d_lcp, bits)
    for p in (d_text, d_sa, d_isa, d_lcp):
        ctx.free(p)
    return err


def test_device_checker_detects_errors(ctx):
    text = inputs.dna(1 << 20, 3)
    assert _device_run_and_check(ctx, text, 32) == [0, 0, 0, 0]
    err = _device_run_and_check(ctx, text, 32, corrupt=True)
    assert err[2] == 1 and err[0] == 0 and err[1] == 0


def test_full_size_config_c2(ctx):
    # BASELINE.json configs[1] at full size: 256 MiB random DNA, uint32; verified on the device by the
    # order property and direct LCP comparison (size-independent properties)
    text = inputs.dna(1 << 28, 1)
    assert _device_run_and_check(ctx, text, 32) == [0, 0, 0, 0]


def test_large_uint64(ctx):
    text = inputs.ascii128((1 << 26) + 3, 42)
    assert _device_run_and_check(ctx, text, 64) == [0, 0, 0, 0]


def test_reduced_memory_layout(ctx, monkeypatch):
    # the large-n layout (output buffers double as sort scratch), forced at a small size
    monkeypatch.setenv("PSACX_FORCE_DIET", "1")
    for bits in (32, 64):
        for text in (O.rand_dna(300007, 5), inputs.ascii128(200003, 3), inputs.dna((1 << 22) + 77, 8)):
            got = run(ctx, text, bits=bits)
            if text.size < (1 << 21):
                ref = O.construct(text, bits=bits)
                assert np.array_equal(got.local_SA, ref["SA"]) and np.array_equal(got.local_B, ref["ISA"])
                assert np.array_equal(got.local_LCP, ref["LCP"])
            else:
                assert O.check_sa(text, got.local_SA, got.local_B) == 0
                assert np.array_equal(O.kasai(text, got.local_SA, got.local_B), got.local_LCP)
    # no-LCP variant and a capacity overflow that must fail loudly, not corrupt
    got = run(ctx, O.rand_dna(100003, 9), bits=32, lcp=False)
    assert np.array_equal(got.local_SA, O.construct(O.rand_dna(100003, 9), bits=32, lcp=False)["SA"])
    monkeypatch.setenv("PSACX_DIET_CAP", "2000")
    import psac_amd
    with pytest.raises(psac_amd.PsacxError):
        run(ctx, inputs.tandem(100000, 64, O.rand_dna(64, 1)), bits=32)


def test_reduced_memory_layout_refines_in_slabs(ctx, monkeypatch):
    # more unresolved suffixes than the reduced-memory layout has room for: the refinement rounds work through
    # slabs of whole buckets (construct.hpp), and a two-stage first round that meets more ties than fit is
    # run again as one sort.  Final SA / ISA / LCP are unique, so they must equal the oracle's bit for bit.
    monkeypatch.setenv("PSACX_FORCE_DIET", "1")
    monkeypatch.setenv("PSACX_DIET_CAP", "40000")
    for bits in (32, 64):
        text = inputs.tandem(300000, 64, O.rand_dna(64, 1))
        got = run(ctx, text, bits=bits)
        ref = O.construct(text, bits=bits)
        assert np.array_equal(got.local_SA, ref["SA"]) and np.array_equal(got.local_B, ref["ISA"]) and np.array_equal(got.local_LCP, ref["LCP"])
        assert len(got.rounds) >= 10
    # low-entropy text: buckets of very different sizes, some slabs hold thousands of buckets
    rng = np.random.RandomState(11)
    p = 0.5 ** np.arange(1, 9); p /= p.sum()
    text = (97 + rng.choice(8, size=400003, p=p)).astype(np.uint8)
    monkeypatch.setenv("PSACX_DIET_CAP", "60000")
    got = run(ctx, text, bits=32)
    ref = O.construct(text, bits=32)
    assert np.array_equal(got.local_SA, ref["SA"]) and np.array_equal(got.local_B, ref["ISA"]) and np.array_equal(got.local_LCP, ref["LCP"])
    # above the two-stage threshold (n >= 2^21): every suffix ties on the leading bits -> one-stage retry -> slabs
    monkeypatch.setenv("PSACX_DIET_CAP", str(1 << 19))
    text = inputs.tandem((1 << 21) + 5, 1024, inputs.dna(1024, 3))
    for bits in (32, 64):
        got = run(ctx, text, bits=bits)
        ref = O.construct(text, bits=bits)
        assert np.array_equal(got.local_SA, ref["SA"]) and np.array_equal(got.local_B, ref["ISA"]) and np.array_equal(got.local_LCP, ref["LCP"])


def test_cpp_header_program(ctx, tmp_path):
    # include/suffix_array.hpp used from C++11 like the reference's class (tests/cpp/test_header.cpp)
    import subprocess
    root = os.path.dirname(HERE)
    exe = str(tmp_path / "test_header")
    lib = os.path.join(root, "psac_amd", "lib")
    cmd = ["g++", "-std=c++11", "-O1", "-Wall", "-o", exe, os.path.join(HERE, "cpp", "test_header.cpp"),
           "-L" + lib, "-lpsacx", "-Wl,-rpath," + lib, "-Wl,-rpath,/opt/rocm/lib"]
    b = subprocess.run(cmd, capture_output=True, text=True)
    assert b.returncode == 0, b.stderr
    r = subprocess.run([exe, str(tmp_path)], capture_output=True, text=True)
    assert r.returncode == 0 and "cpp header tests passed" in r.stdout, r.stdout + r.stderr


def test_suffix_tree_nodes(ctx):
    # psac -t: construct_suffix_tree (suffix_tree.hpp:413-499).  Known answer for mississippi
    # (test/test_suffixtree.cpp:68-83), oracle for random DNA (:89-122) and (abc)^n repeats (:126-162)
    import psac_amd
    m = KAT["mississippi"]
    r = O.construct(m["text"], bits=64)
    nodes = psac_amd.suffix_tree(m["text"], r["SA"], r["LCP"], ctx=ctx)
    assert nodes.reshape(-1).tolist() == m["suffix_tree_nodes"]
    for text, bits in ((O.rand_dna(116, 13), 64), (O.rand_dna(1000, 13), 32), (O.rand_dna(23713, 13), 64),
                       (inputs.cyclic(3000, "abc"), 32), (inputs.ascii128(5000, 2), 64)):
        sa = run(ctx, text, bits=bits)
        got = psac_amd.suffix_tree(text, sa.local_SA, sa.local_LCP, ctx=ctx)
        assert np.array_equal(got, O.suffix_tree(text, sa.local_SA, sa.local_LCP))


def test_left_branching_chars(ctx):
    # suffix_array<char, T, true, true> (suffix_array.hpp:170, :211-212, :1365-1383; par_rmq.hpp:334-481):
    # the engine's Lc against the oracle's (carried through the leftmost range minima) and the definition
    import psac_amd
    cases = [(O.as_text("mississippi"), 64, True, 0), (O.rand_dna(130370, 7), 32, True, 0), (O.rand_dna(66763, 23), 64, False, 2),
             (inputs.cyclic(39999, "abc"), 32, True, 0), (inputs.tandem(100000, 1024, inputs.dna(1024, 5)), 32, True, 0),
             (inputs.ascii128(200000, 9), 64, True, 0), (O.as_text("aaaaaaaaaaaaaaaa"), 32, True, 0), (O.as_text("ab"), 32, True, 0),
             (O.as_text("a"), 64, True, 0)]
    for text, bits, fast, k in cases:
        sa = psac_amd.SuffixArray(index_bits=bits, lc=True, ctx=ctx)
        sa.construct(text, fast_resolval=fast, k=k)
        if text.size > 1:
            ref = O.construct_lc(text, bits=bits, fast=fast, k=k)
            assert np.array_equal(sa.local_SA, ref["SA"]) and np.array_equal(sa.local_LCP, ref["LCP"])
            assert np.array_equal(sa.local_Lc, ref["Lc"])
        assert np.array_equal(sa.local_Lc, O.left_chars_by_definition(text, sa.local_SA, sa.local_LCP))
    # 2^24 DNA: definition only (size-independent property)
    text = inputs.dna(1 << 24, 1)
    sa = psac_amd.SuffixArray(index_bits=32, lc=True, ctx=ctx)
    sa.construct(text)
    assert np.array_equal(sa.local_Lc, O.left_chars_by_definition(text, sa.local_SA, sa.local_LCP))
    # Lc without LCP is refused
    import ctypes as C
    buf = np.zeros(16, np.uint32); lc = np.zeros(16, np.uint8); t = O.as_text("abracadabraabrac")
    rc = ctx._lib.psacx_construct_lc_u32(ctx.handle, t.ctypes.data, 16, 0, 0, buf.ctypes.data, buf.ctypes.data, None, lc.ctypes.data)
    assert rc == -1


def test_benchmark_clis(tmp_path):
    # src/benchmark.cpp:35-80 ("p;method;ms"), src/benchmark_k.cpp:35-67 ("p;method;k;ms"),
    # src/benchmark_ansv.cpp ("n;p;method;ms"): same flags and CSV columns
    import subprocess
    root = os.path.dirname(HERE)
    bdir = os.path.join(root, "psac_amd", "bin")
    sac, bk, ba = (os.path.join(bdir, x) for x in ("benchmark_sac", "benchmark_k", "benchmark-ansv"))
    if not all(os.path.exists(x) for x in (sac, bk, ba)):
        pytest.skip("benchmark CLIs not built")
    r = subprocess.run([sac, "-r", "200000", "-i", "2"], capture_output=True, text=True)
    assert r.returncode == 0, r.stderr
    rows = [l.split(";") for l in r.stdout.split()]
    assert [x[1] for x in rows] == ["reg-nolcp", "reg-fast-nolcp", "reg-lcp", "reg-fast-lcp"] * 2
    assert all(x[0] == "1" and float(x[2]) > 0 for x in rows)
    f = tmp_path / "t.txt"
    f.write_bytes(bytes(inputs.dna(50000, 3)))
    r = subprocess.run([bk, "-f", str(f), "-k", "4"], capture_output=True, text=True)
    assert r.returncode == 0, r.stderr
    rows = [l.split(";") for l in r.stdout.split()]
    assert [(x[0], x[1], x[2]) for x in rows] == [("1", "reg-fast-nolcp", "4"), ("1", "reg-nolcp", "4")]
    for flag in ("-u", "-k", "-b"):
        r = subprocess.run([ba, "-n", "100000", flag], capture_output=True, text=True)
        assert r.returncode == 0, r.stderr
        x = r.stdout.split()[0].split(";")
        assert x[0] == "100000" and x[1] == "1" and x[2] == "gansv-hip" and float(x[3]) > 0
    assert subprocess.run([sac], capture_output=True).returncode != 0


def _gsa(ctx, strings, bits, lcp=True, k=0, sep=None):
    import psac_amd
    sa = psac_amd.SuffixArray(index_bits=bits, lcp=lcp, ctx=ctx)
    sa.construct_ss(strings, sep=sep, k=k)
    return sa


def test_psac_vs_dss_command_line(tmp_path):
    # src/psac_vs_dss.cpp:59-119: the engine and libdivsufsort on the same input, -c = sufcheck on both
    import subprocess
    if not O.have_divsufsort():
        pytest.skip("oracle/_ref/libdivsufsort*.so not built")
    from test_oracle_golden import build_dss_tool
    exe = build_dss_tool("psac_vs_dss", tmp_path, with_engine=True)
    r = subprocess.run([exe, "-r", "500000", "-s", "5", "-c"], capture_output=True, text=True)
    assert r.returncode == 0, r.stderr
    assert "PSAC time:" in r.stderr and "divsufsort time:" in r.stderr and "[SUCCESS]" in r.stderr
    f = tmp_path / "t.txt"
    f.write_bytes(bytes(inputs.ascii128(300001, 5)))
    r = subprocess.run([exe, "-f", str(f), "-c"], capture_output=True, text=True)
    assert r.returncode == 0 and "[SUCCESS]" in r.stderr, r.stderr
    assert subprocess.run([exe], capture_output=True).returncode != 0


def test_gsa_reference_vectors(ctx):
    # test/test_gsa.cpp:73-105 (SimpleTiny) and :107-179 (IncRepeats*): the reference's expected arrays
    from test_oracle_golden import GSA_REPEATS, repeat_inc_gsa, repeat_inc_glcp, repeat_inc_seq
    for bits in (64, 32):
        sa = _gsa(ctx, ["abab", "baba"], bits)
        assert sa.local_SA.tolist() == [7, 2, 5, 0, 3, 6, 1, 4]
        assert sa.local_LCP.tolist() == [0, 1, 2, 3, 0, 1, 2, 3]
        for seq, reps in GSA_REPEATS:
            sa = _gsa(ctx, repeat_inc_seq(seq, reps), bits)
            assert sa.local_SA.tolist() == repeat_inc_gsa(len(seq), reps), (seq, reps, bits)
            assert sa.local_LCP.tolist() == repeat_inc_glcp(len(seq), reps), (seq, reps, bits)
            assert np.array_equal(sa.local_B[sa.local_SA.astype(np.int64)], np.arange(sa.n, dtype=sa.dtype))


def test_gsa_device_entry_point_and_offset_validation(ctx):
    # psacx_construct_gsa_dev_*: text and offsets resident in HBM; malformed offsets (not starting at 0, not ending at n,
    # an empty string) must give PSACX_EINVAL instead of out-of-bounds reads (stringset.hpp:53-72)
    import ctypes as C
    strings = [b"ACGTACGT", b"GATTACA", b"ACG", b"TTTTTTTTTT"]
    text = np.frombuffer(b"".join(strings), np.uint8)
    n = text.size
    off = np.zeros(len(strings) + 1, np.uint64); off[1:] = np.cumsum([len(s) for s in strings])
    d_text = ctx.alloc(n); ctx.h2d(d_text, text)
    d_off = ctx.alloc(off.nbytes)
    d_sa, d_isa, d_lcp = ctx.alloc(n * 8), ctx.alloc(n * 8), ctx.alloc(n * 8)
    fn = ctx._lib.psacx_construct_gsa_dev_u64

    def call(o):
        ctx.h2d(d_off, o)
        return fn(ctx.handle, C.c_void_p(d_text), n, C.c_void_p(d_off), len(strings), 0, 1, C.c_void_p(d_sa), C.c_void_p(d_isa), C.c_void_p(d_lcp))
    assert call(off) == 0
    sa = np.empty(n, np.uint64); lcp = np.empty(n, np.uint64)
    ctx.d2h(sa, d_sa); ctx.d2h(lcp, d_lcp)
    ref = O.construct_ss(strings, bits=64)
    assert np.array_equal(sa, ref["SA"]) and np.array_equal(lcp, ref["LCP"])
    bad1 = off.copy(); bad1[0] = 1
    bad2 = off.copy(); bad2[-1] = n - 1
    bad3 = off.copy(); bad3[2] = bad3[1]
    bad4 = off.copy(); bad4[2], bad4[3] = off[3], off[2]
    for bad in (bad1, bad2, bad3, bad4):
        assert call(bad) == -1
    for p in (d_text, d_off, d_sa, d_isa, d_lcp):
        ctx.free(p)


def test_gsa_against_oracle(ctx):
    rng = np.random.RandomState(11)
    sets = []
    for sigma, m, lo, hi in ((4, 300, 1, 400), (2, 50, 1, 30), (1, 40, 1, 100), (26, 2000, 5, 60), (4, 1, 5000, 5001),
                             (4, 3000, 1, 3), (90, 200, 100, 2000)):
        sets.append([bytes(rng.randint(65, 65 + sigma, size=int(rng.randint(lo, hi))).astype(np.uint8)) for _ in range(m)])
    # many copies of the same reads (deep ties) and reads that are prefixes of each other
    base = bytes(inputs.dna(300, 4))
    sets.append([base[:int(x)] for x in rng.randint(1, 300, size=500)])
    sets.append([base] * 200)
    for strings in sets:
        for bits, k in ((32, 0), (64, 0), (32, 3)):
            got = _gsa(ctx, strings, bits, k=k)
            ref = O.construct_ss(strings, bits=bits, k=k)
            assert np.array_equal(got.local_SA, ref["SA"]), (bits, k)
            assert np.array_equal(got.local_B, ref["ISA"]), (bits, k)
            assert np.array_equal(got.local_LCP, ref["LCP"]), (bits, k)
        nol = _gsa(ctx, strings, 32, lcp=False)
        assert np.array_equal(nol.local_SA, ref["SA"].astype(np.uint32))
    # flat buffer with separator runs (stringset.hpp:43-72), the way gsac reads a file (src/gsac.cpp:169-170)
    flat = b"\n\n" + b"\n".join(sets[0]) + b"\n\n\n" + b"\n".join(sets[3]) + b"\n"
    got = _gsa(ctx, flat, 64, sep="\n")
    ref = O.construct_ss(sets[0] + sets[3], bits=64)
    assert np.array_equal(got.local_SA, ref["SA"]) and np.array_equal(got.local_LCP, ref["LCP"])


def test_gsa_large_properties(ctx):
    # 2^22 characters in ~40k reads: sortedness and LCP by direct comparison of neighbours on a sample,
    # SA a permutation, ISA its inverse
    rng = np.random.RandomState(3)
    text = inputs.dna(1 << 22, 9)
    cuts = np.unique(np.concatenate([[0, text.size], rng.randint(1, text.size, size=40000)]))
    strings = [bytes(text[a:b]) for a, b in zip(cuts[:-1], cuts[1:])]
    sa = _gsa(ctx, strings, 32)
    n = text.size
    SA = sa.local_SA.astype(np.int64)
    assert np.array_equal(np.sort(SA), np.arange(n))
    assert np.array_equal(sa.local_B[SA], np.arange(n, dtype=np.uint32))
    ends = cuts[np.searchsorted(cuts, SA, side="right")]
    for i in rng.randint(1, n, size=20000):
        a = bytes(text[SA[i - 1]:ends[i - 1]]); b = bytes(text[SA[i]:ends[i]])
        assert a < b or (a == b and SA[i - 1] < SA[i])
        c = 0
        while c < len(a) and c < len(b) and a[c] == b[c]:
            c += 1
        assert sa.local_LCP[i] == c


def test_gsac_cli(tmp_path):
    # src/gsac.cpp:139-204: gsac -f <file> [-l] [-c]; strings are the lines of the file
    import subprocess
    root = os.path.dirname(HERE)
    gsac = os.path.join(root, "psac_amd", "bin", "gsac")
    if not os.path.exists(gsac):
        pytest.skip("gsac not built")
    rng = np.random.RandomState(2)
    strings = [bytes(rng.randint(65, 69, size=int(rng.randint(1, 200))).astype(np.uint8)) for _ in range(500)]
    f = tmp_path / "reads.txt"
    f.write_bytes(b"\n".join(strings) + b"\n")
    r = subprocess.run([gsac, "-f", str(f), "-l", "-c", "-o", str(tmp_path / "g")], capture_output=True, text=True)
    assert r.returncode == 0 and "[SUCCESS] GSA correct" in r.stdout and "PSAC time:" in r.stderr, r.stdout + r.stderr
    ref = O.construct_ss(strings, bits=64)
    assert np.array_equal(np.fromfile(str(tmp_path / "g.sa64"), dtype=np.uint64), ref["SA"])
    assert np.array_equal(np.fromfile(str(tmp_path / "g.lcp64"), dtype=np.uint64), ref["LCP"])
    r = subprocess.run([gsac, "-f", str(f), "-c"], capture_output=True, text=True)
    assert r.returncode == 0 and "[SUCCESS]" in r.stdout
    assert subprocess.run([gsac], capture_output=True).returncode != 0
    # the same string set on three ranks (here sharing device 0): construct_ss on p ranks, suffix_array.hpp:267-363
    r = subprocess.run([gsac, "-f", str(f), "-l", "-c", "-o", str(tmp_path / "g3"), "--gpus-on-device", "0,3"], capture_output=True, text=True)
    assert r.returncode == 0 and "[SUCCESS] GSA correct" in r.stdout, r.stdout + r.stderr
    assert np.array_equal(np.fromfile(str(tmp_path / "g3.sa64"), dtype=np.uint64), ref["SA"])
    assert np.array_equal(np.fromfile(str(tmp_path / "g3.lcp64"), dtype=np.uint64), ref["LCP"])


def test_distributed_ansv_on_gpu(ctx):
    # the HIP search op (psacx_op_nsv_from_*) under the loopback world: P virtual ranks sharing this GPU
    import torch
    from dist_harness import dist as D
    from dist_harness.comm import LoopbackWorld
    from dist_harness.dist_ops import HipOps
    rng = np.random.RandomState(8)
    text = inputs.dna(200000, 6)
    lcp = run(ctx, text, bits=32).local_LCP
    cases = [(rng.randint(0, 4, size=5000).astype(np.uint64), 32), (rng.randint(0, 10**6, size=70000).astype(np.uint64), 64),
             (lcp.astype(np.uint64), 32), (np.zeros(3000, np.uint64), 64), (np.arange(5000, dtype=np.uint64), 32)]
    for vals, bits in cases:
        v = vals.astype(np.uint32 if bits == 32 else np.uint64)
        none = (1 << bits) - 1
        for P in (1, 2, 4):
            sizes = D.blk_sizes(vals.size, P)
            offs = D.prefix(sizes)
            ops = [HipOps(bits, 0) for _ in range(P)]
            sdt = np.int32 if bits == 32 else np.int64
            blocks = [torch.from_numpy(v[o:o + s].view(sdt).copy()).cuda() for o, s in zip(offs, sizes)]
            for lt, rt in ((0, 0), (2, 0), (1, 2), (2, 1)):
                def fn(comm, op, blk):
                    return (yield from D.dist_ansv(comm, op, blk, lt, rt))
                res = LoopbackWorld(P).run(fn, [(ops[r], blocks[r]) for r in range(P)])
                udt = np.uint32 if bits == 32 else np.uint64
                L = np.concatenate([x[0].cpu().numpy().view(udt) for x in res]).astype(np.uint64)
                R = np.concatenate([x[1].cpu().numpy().view(udt) for x in res]).astype(np.uint64)
                assert np.array_equal(L, O.ansv(v, True, lt, none)), (bits, P, lt)
                assert np.array_equal(R, O.ansv(v, False, rt, none)), (bits, P, rt)
            for o in ops:
                o.close()


def test_python_cli_single_and_torchrun(tmp_path):
    # `python -m psac_amd` = src/psac.cpp's command line for the block-distributed path; with one process it
    # uses the single-GPU engine, under torchrun (here: one rank, RCCL) the distributed choreography
    import subprocess
    root = os.path.dirname(HERE)
    text = inputs.dna(300000, 12)
    f = tmp_path / "t.txt"
    f.write_bytes(bytes(text))
    ref = O.construct(text, bits=32)
    env = dict(os.environ, PYTHONPATH=root)
    r = subprocess.run([sys.executable, "-m", "psac_amd", "-f", str(f), "-l", "-c", "-o", str(tmp_path / "a")],
                       capture_output=True, text=True, env=env, cwd=root)
    assert r.returncode == 0 and "[SUCCESS]" in r.stderr and "PSAC time:" in r.stderr, r.stderr[-2000:]
    assert np.array_equal(np.fromfile(str(tmp_path / "a.sa64"), np.uint64), ref["SA"].astype(np.uint64))
    assert np.array_equal(np.fromfile(str(tmp_path / "a.lcp64"), np.uint64), ref["LCP"].astype(np.uint64))
    r = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "1", "--master-addr",
                        "127.0.0.1", "--master-port", "29577", "-m", "psac_amd", "-f", str(f), "-l", "-o", str(tmp_path / "b")],
                       capture_output=True, text=True, env=dict(env, PSACX_CLI_FORCE_DIST="1"), cwd=root)
    assert r.returncode == 0 and "PSAC time:" in r.stderr, r.stderr[-2000:]
    assert np.array_equal(np.fromfile(str(tmp_path / "b.sa64"), np.uint64), ref["SA"].astype(np.uint64))
    assert np.array_equal(np.fromfile(str(tmp_path / "b.lcp64"), np.uint64), ref["LCP"].astype(np.uint64))
    # -r draws the reference's generator (alphabet.hpp:32-45)
    r = subprocess.run([sys.executable, "-m", "psac_amd", "-r", "20000", "-s", "0", "-c"], capture_output=True, text=True, env=env, cwd=root)
    assert r.returncode == 0 and "[SUCCESS]" in r.stderr, r.stderr[-2000:]


def test_degenerate_texts_above_the_two_stage_threshold(ctx):
    # n >= 2^21 switches on the two-stage first round and the three-kernel radix passes: single-symbol and
    # period-2 texts (every suffix ties, 15-18 rounds), a lone different last character, and n = 2^21 exactly
    # (the only uint32 size where stage 1 leaves low bits of word 1 unsorted)
    a = np.full((1 << 21) + 3, 65, np.uint8)
    cases = [(a, 32), (inputs.cyclic(1 << 22, "AB"), 32), (a[:1 << 21], 64),
             (np.concatenate([np.full((1 << 22) - 1, 65, np.uint8), np.array([66], np.uint8)]), 32), (inputs.dna(1 << 21, 3), 32)]
    for text, bits in cases:
        same_as_oracle(ctx, text, bits=bits)


def test_one_word_form_of_the_prefix_sort(ctx, monkeypatch):
    # 64-bit words, at most 2^32 characters: the prefix sort of the first round partitions word 1 by the TOP digit of the prefix
    # and moves one 64-bit word per record (rest of the prefix | suffix) through LSD passes inside the 256 buckets
    # (engine.hpp: prefix_sort_1w; from 2^24 characters on by default, from 2^21 here).  Random DNA and ASCII (all buckets in
    # use, ragged last tiles), alphabets of 3 and 5 symbols (most buckets empty, the others uneven), a text whose tie groups are
    # long (radix fallback of the ties with word 1 read from the text again), a single symbol (one bucket), the reduced-memory
    # layout, a caller-supplied k -- and the same results with the form switched off.
    monkeypatch.setenv("PSACX_ONE_WORD_MIN", "21")
    rng = np.random.RandomState(12)
    three = np.frombuffer(b"ACG", np.uint8)[rng.randint(0, 3, (1 << 22) + 5)].copy()
    five = np.frombuffer(b"ACGNT", np.uint8)[rng.choice(5, (1 << 21) + 4099, p=[0.3, 0.2, 0.2, 0.01, 0.29])].copy()
    rep = np.tile(inputs.dna(1 << 12, 9), 1 << 10)
    rep[::4099] = 84
    cases = [(inputs.dna((1 << 22) + 77, 5), {}), (inputs.ascii128((1 << 21) + 5, 4), {}), (three, {}), (five, {}), (rep, {}),
             (np.full((1 << 21) + 3, 65, np.uint8), {}), (inputs.dna((1 << 21) + 100, 17), {"k": 12})]
    for text, kw in cases:
        got, ref = same_as_oracle(ctx, text, bits=64, **kw)
        assert [(h, b, e) for (h, b, e, *_rest) in got.rounds] == [(h, b, e) for h, b, e, _ in ref["trace"]]
    # (texts whose sampled prefixes nearly all repeat -- `rep` and the single symbol above -- take one sort over both words instead:
    #  prefix_dup_probe_kernel, PSACX_RETRY_1STAGE; a text of which a third repeats keeps two stages in two-array passes, PSACX_RETRY_1W)
    third = inputs.dna((1 << 22) + 9, 31)
    third[: (1 << 22) // 3] = np.tile(inputs.dna(1 << 11, 33), (1 << 22) // 3 // (1 << 11) + 1)[: (1 << 22) // 3]
    got, ref = same_as_oracle(ctx, third, bits=64)
    assert [(h, b, e) for (h, b, e, *_rest) in got.rounds] == [(h, b, e) for h, b, e, _ in ref["trace"]]
    # ... and the one-word form forced on the repetitive ones
    monkeypatch.setenv("PSACX_ONE_WORD_ALWAYS", "1")
    same_as_oracle(ctx, rep, bits=64)
    same_as_oracle(ctx, np.full((1 << 21) + 3, 65, np.uint8), bits=64)
    monkeypatch.delenv("PSACX_ONE_WORD_ALWAYS")
    monkeypatch.setenv("PSACX_FORCE_DIET", "1")
    same_as_oracle(ctx, inputs.dna((1 << 22) + 1, 6), bits=64)
    same_as_oracle(ctx, five, bits=64)
    monkeypatch.delenv("PSACX_FORCE_DIET")
    monkeypatch.setenv("PSACX_NO_ONE_WORD", "1")
    same_as_oracle(ctx, five, bits=64)
    monkeypatch.delenv("PSACX_NO_ONE_WORD")
    # from 2^22 characters on the tie stage and the rebucket kernel read the one-word records where the sort left them (the cases of
    # 2^22 characters and more above; `rep` with the probe off meets a long tie group and widens them after all); the same with the
    # last pass writing word 1 and the suffixes as arrays
    monkeypatch.setenv("PSACX_WIDEN_LAST", "1")
    same_as_oracle(ctx, cases[0][0], bits=64)
    same_as_oracle(ctx, three, bits=64)


def test_bucket_ids_of_resolved_tiles_are_filled_in_on_demand(ctx, monkeypatch):
    # rebucket_first_kernel (fused one-GPU form, from 2^22 characters on) does not write the bucket ids of a tile without
    # unresolved suffixes; when some OTHER tile has unresolved suffixes, run_compact fills them in before anybody reads them
    # (fill_resolved_ids_kernel).  Random DNA with one long repeat: two stretches of SA stay unresolved for several rounds, the
    # rest of the tiles are resolved after the first.  Both word sizes, the per-round log included.
    text = inputs.dna((1 << 22) + 1000, 23)
    text[3000000:3050000] = text[100000:150000]
    for bits in (64, 32):
        got, ref = same_as_oracle(ctx, text, bits=bits)
        assert len(ref["trace"]) > 3
        assert [(h, b, e) for (h, b, e, *_rest) in got.rounds] == [(h, b, e) for h, b, e, _ in ref["trace"]]


def test_refinement_round_forms(ctx):
    # rounds with at least 7/8 of the suffixes unresolved take all n records in text order and rebuild ISA by inverting SA
    # (shift_keys_kernel; not when SA order is nearly text order: sa_locality_kernel); 64-bit words below 2^32 characters
    # sort two-word records (bucket id and rank h further in one word, 32-bit suffix) from 2^21 records on.  Every form
    # must give the oracle's arrays and per-round log (the single symbol takes the list form, the repeats the whole rounds; 32-bit
    # words and the texts below 2^21 unresolved suffixes the three-word records).
    texts = [inputs.tandem((1 << 21) + 3000, 512, inputs.dna(512, 3)), np.full((1 << 21) + 17, 67, np.uint8),
             np.tile(inputs.dna(1 << 11, 9), (1 << 10) + 1)]
    texts[2][::4099] = 84
    for text in texts:
        for bits in (64, 32):
            got, ref = same_as_oracle(ctx, text, bits=bits)
            assert [(h, b, e) for (h, b, e, *_rest) in got.rounds] == [(h, b, e) for h, b, e, _ in ref["trace"]]


def test_refinement_sort_inside_lds(ctx, monkeypatch):
    # bucket_sort.hpp: 64-bit words, a list round of at least 2^21 unresolved suffixes none of whose buckets is longer than a workgroup
    # holds in LDS -- every bucket is sorted there (one pass reported for the round) instead of by the global radix passes.  Repeated reads
    # with mutations: 2048 copies of 4096 characters, so no bucket outgrows 2048 suffixes.  Both forms must give the oracle's arrays and log.
    text = inputs.mutated((1 << 23) + 1234, 4096, 5)
    got, ref = same_as_oracle(ctx, text, bits=64)
    assert [(h, b, e) for (h, b, e, *_rest) in got.rounds] == [(h, b, e) for h, b, e, _ in ref["trace"]]
    in_lds = [r for r in got.rounds[1:] if r[3] >= (1 << 21) and r[4] == 1]
    assert in_lds, got.rounds
    monkeypatch.setenv("PSACX_NO_BUCKET_SORT", "1")
    other, _ = same_as_oracle(ctx, text, bits=64)
    assert not [r for r in other.rounds[1:] if r[3] >= (1 << 21) and r[4] == 1]
    assert np.array_equal(other.local_SA, got.local_SA) and np.array_equal(other.local_LCP, got.local_LCP)


def test_digit_bytes_between_the_three_kernel_sort_passes(ctx, monkeypatch):
    # engine.hpp: dispatch_pass3 -- the sorts of 64-bit words with 32-bit payloads (the one-stage first sort of a repetitive text over both key
    # words, the text-order rounds, the list rounds when the LDS sort is switched off) leave the next pass's digit as a byte beside the records
    # and take their tile histograms from it (radix_tile_hist_bytes_flat_kernel: runs of one digit, the keys of repeated reads, add once per
    # segment of lanes).  With and without the bytes: the oracle's arrays, the same log of rounds and passes.
    import psac_amd
    monkeypatch.setattr(psac_amd._lib, "ENV_KNOBS", False)
    texts = [inputs.mutated((1 << 23) + 1234, 4096, 5), inputs.tandem((1 << 22) + 77, 1024, inputs.dna(1024, 3)),
             np.concatenate([inputs.dna(1 << 21, 9), inputs.mutated(1 << 22, 512, 6), inputs.dna(3001, 10)])]
    def same(x, ref):
        return np.array_equal(x.local_SA, ref["SA"]) and np.array_equal(x.local_B, ref["ISA"]) and np.array_equal(x.local_LCP, ref["LCP"])
    try:
        for text in texts:
            ctx.configure(reset=0)
            a, ref = same_as_oracle(ctx, text, bits=64)
            assert [(h, x, e) for (h, x, e, *_r) in a.rounds] == [(h, x, e) for h, x, e, _ in ref["trace"]]
            ctx.configure(no_bucket_sort=1)
            b = run(ctx, text, bits=64)
            ctx.configure(no_bucket_sort=1, no_digit_bytes=1)
            c = run(ctx, text, bits=64)
            assert same(b, ref) and same(c, ref) and b.rounds == c.rounds
            ctx.configure(reset=0); ctx.configure(no_digit_bytes=1)
            d = run(ctx, text, bits=64)
            assert same(d, ref) and d.rounds == a.rounds
    finally:
        ctx.configure(reset=0)


def test_options_through_the_abi(ctx, monkeypatch):
    # psacx_configure (include/psacx.h): the forms of single stages are options of the context; the library itself never reads the
    # environment (the suite's PSACX_* variables go through the debug shim psacx_configure_from_env, switched off here)
    import psac_amd
    monkeypatch.setattr(psac_amd._lib, "ENV_KNOBS", False)
    monkeypatch.setenv("PSACX_NO_BUCKET_SORT", "1")               # (must be ignored now)
    text = inputs.mutated((1 << 23) + 1234, 4096, 5)          # (the text of test_refinement_sort_inside_lds)
    in_lds = lambda sa: [r for r in sa.rounds[1:] if r[3] >= (1 << 21) and r[4] == 1]
    try:
        ctx.configure(reset=0)                    # (whatever the shim set for the test before this one)
        a = run(ctx, text, bits=64)
        assert in_lds(a), a.rounds
        ctx.configure(no_bucket_sort=1)
        b = run(ctx, text, bits=64)
        assert not in_lds(b) and np.array_equal(a.local_SA, b.local_SA) and np.array_equal(a.local_LCP, b.local_LCP)
        ctx.configure(no_bucket_sort=0, force_diet=1, diet_cap=1 << 20)
        c = run(ctx, text, bits=64)
        assert np.array_equal(a.local_SA, c.local_SA) and np.array_equal(a.local_B, c.local_B) and np.array_equal(a.local_LCP, c.local_LCP)
        ctx.configure(reset=0)
        d = run(ctx, text, bits=64)
        assert in_lds(d) and d.rounds == a.rounds
        assert ctx._lib.psacx_configure(ctx.handle, 999, 1) == -1 and ctx._lib.psacx_configure(ctx.handle, psac_amd._lib.OPTIONS["gather"], 7) == -1
        # ... and the shim: the same variable, asked for explicitly
        ctx.check(ctx._lib.psacx_configure_from_env(ctx.handle))
        e = run(ctx, text, bits=64)
        assert not in_lds(e)
    finally:
        ctx.configure(reset=0)


def _construct_dev64(ctx, text):
    import psac_amd
    n = int(text.size)
    d_text = ctx.alloc(n); ctx.h2d(d_text, text)
    d = [ctx.alloc(n * 8) for _ in range(3)]
    sa = psac_amd.SuffixArray(index_bits=64, lcp=True, ctx=ctx)
    st = sa.construct_device(d_text, n, d[0], d[1], d[2])
    out = [np.empty(n, np.uint64) for _ in range(3)]
    for a, p in zip(out, d):
        ctx.d2h(a, p)
    for p in [d_text] + d:
        ctx.free(p)
    return out, st, sa.rounds


def test_long_bucket_rounds_through_levels_and_the_heavy_light_split(ctx, monkeypatch):
    # Rounds whose buckets are too long for the sort in LDS (a tandem repeat: period-many buckets of n / period suffixes): the ranks h
    # further come through partition levels by text position (construct.hpp: gather_by_levels, psac's bulk_rma batched by owner,
    # bulk_rma.hpp:20-49) and, with at most 4096 buckets, the records that carry their bucket's heavy rank skip the radix sort
    # (heavy_keys.hpp).  Every form -- one random fetch per record and a sort of all records (the older path), levels without the split,
    # both layouts, a round in slabs -- must give the same SA / ISA / LCP and the same round log; one text also against the oracle.
    n = (1 << 24) + 4321
    rng = np.random.RandomState(11)
    spotted = inputs.tandem(n, 1024, inputs.dna(1024, 3))
    spotted[rng.randint(0, n, 40)] = 84                        # a few substitutions: buckets split early, many light records
    texts = [inputs.tandem(n, 1024, inputs.dna(1024, 3)), inputs.tandem(n, 96, inputs.dna(96, 5)), spotted]
    for ti, text in enumerate(texts):
        for k_ in ("PSACX_GATHER", "PSACX_NO_HEAVY", "PSACX_FORCE_DIET", "PSACX_DIET_CAP"):
            monkeypatch.delenv(k_, raising=False)
        (SA, ISA, LCP), st, log = _construct_dev64(ctx, text)
        assert st.heavy_rounds > 0 and st.level_gathers >= st.heavy_rounds and st.heavy_records > st.light_records, (st.heavy_rounds, st.level_gathers)
        if ti == 0:
            oSA, oLCP = O.construct_all_cores(text, bits=64)
            assert np.array_equal(SA, oSA) and np.array_equal(LCP, oLCP)
            assert np.array_equal(ISA[SA.astype(np.int64)], np.arange(n, dtype=np.uint64))
        forms = [{"PSACX_GATHER": "fetch"}, {"PSACX_NO_HEAVY": "1"}, {"PSACX_FORCE_DIET": "1"},
                 {"PSACX_FORCE_DIET": "1", "PSACX_DIET_CAP": str(6 << 20)}, {"PSACX_FORCE_DIET": "1", "PSACX_NO_HEAVY": "1"}]
        for fi, env in enumerate(forms if ti != 1 else forms[:3]):
            for k_ in ("PSACX_GATHER", "PSACX_NO_HEAVY", "PSACX_FORCE_DIET", "PSACX_DIET_CAP"):
                monkeypatch.delenv(k_, raising=False)
            for k_, v_ in env.items():
                monkeypatch.setenv(k_, v_)
            (a, b, c), st2, log2 = _construct_dev64(ctx, text)
            assert np.array_equal(a, SA) and np.array_equal(b, ISA) and np.array_equal(c, LCP), (ti, env)
            if "PSACX_DIET_CAP" not in env:             # (the counters of a round in slabs may run ahead of the one-step log)
                assert [r[:3] for r in log2] == [r[:3] for r in log], (ti, env)
            if env.get("PSACX_GATHER") == "fetch":
                assert st2.level_gathers == 0 and st2.heavy_rounds == 0
            elif "PSACX_NO_HEAVY" in env:
                assert st2.level_gathers > 0 and st2.heavy_rounds == 0
            else:
                assert st2.heavy_rounds > 0, (ti, env)


def test_ansv_device_resident(ctx):
    # psacx_ansv_dev_*: LCP left in HBM by the construction -> ANSV without leaving the device (psac -t's
    # pair: left furthest_eq, right nearest_sm, suffix_tree.hpp:62); 2^24 characters, compared with the oracle
    import psac_amd
    n = 1 << 24
    text = inputs.dna(n, 21)
    d_text = ctx.alloc(n); ctx.h2d(d_text, text)
    d_sa, d_isa, d_lcp = ctx.alloc(n * 4), ctx.alloc(n * 4), ctx.alloc(n * 4)
    d_l, d_r = ctx.alloc(n * 8), ctx.alloc(n * 8)
    sa = psac_amd.SuffixArray(index_bits=32, lcp=True, ctx=ctx)
    sa.construct_device(d_text, n, d_sa, d_isa, d_lcp)
    none = (1 << 64) - 1
    psac_amd.ansv_device(ctx, d_lcp, n, d_l, d_r, 32, 2, 0, none)
    lcp = np.empty(n, np.uint32); L = np.empty(n, np.uint64); R = np.empty(n, np.uint64)
    ctx.d2h(lcp, d_lcp); ctx.d2h(L, d_l); ctx.d2h(R, d_r)
    assert np.array_equal(L, O.ansv(lcp, True, 2, none)) and np.array_equal(R, O.ansv(lcp, False, 0, none))
    for p in (d_text, d_sa, d_isa, d_lcp, d_l, d_r):
        ctx.free(p)


def test_no_fast_and_k_above_the_two_stage_threshold(ctx):
    # fast_resolval = false (every round works on all n records) and a caller-supplied k with the two-stage first
    # round active (n >= 2^21); uint32 and uint64
    text = inputs.dna((1 << 21) + 100, 17)
    same_as_oracle(ctx, text, bits=32, fast=False)
    same_as_oracle(ctx, text, bits=64, fast=False, k=12)
    same_as_oracle(ctx, text, bits=32, k=7)
    text = inputs.ascii128((1 << 21) + 5, 4)
    same_as_oracle(ctx, text, bits=64)
    same_as_oracle(ctx, text, bits=32, fast=False)


def test_scaled_down_twins_of_the_baseline_configs(ctx):
    # every BASELINE config divided by 256 must match the CPU restatement bit for bit.
    #   C3 / 256: 2^24 random ASCII (sigma = 128), uint64, one GPU
    #   C4 / 256: 2^26 random DNA, uint64, 8 ranks of 2^23 (virtual ranks sharing this GPU, the HIP step ops)
    # ... and, independently, libdivsufsort + Kasai (the reference's own check, test/test_psac.cpp:77-98, :50-74)
    text = inputs.ascii128(1 << 24, 42)
    got = run(ctx, text, bits=64)
    SA, LCP = O.construct_all_cores(text, bits=64)
    assert np.array_equal(got.local_SA, SA) and np.array_equal(got.local_LCP, LCP)
    assert np.array_equal(got.local_B[got.local_SA.astype(np.int64)], np.arange(text.size, dtype=np.uint64))
    if O.have_divsufsort():
        dSA, dISA, dLCP = O.divsufsort_sa_lcp(text, 64)
        assert np.array_equal(got.local_SA, dSA) and np.array_equal(got.local_B, dISA) and np.array_equal(got.local_LCP, dLCP)
    text = inputs.dna(1 << 26, 1)
    sa, isa, lcp, _ = _dist_loopback_gpu(text, 8, 64)
    SA, LCP = O.construct_all_cores(text, bits=64)
    assert np.array_equal(sa, SA) and np.array_equal(lcp, LCP)
    assert np.array_equal(isa[sa.astype(np.int64)], np.arange(text.size, dtype=np.uint64))
    if O.have_divsufsort():
        assert np.array_equal(sa, O.divsufsort(text, 64))

@pytest.mark.gpu
def test_ansv_answers_carried_from_tile_to_tile(ctx):
    # 2^25 elements: every wave of the kernel works through eight tiles per side and carries the answers that lie beyond a tile edge
    # from one to the next (psac_amd/csrc/ansv_wave.hpp step 6; arrays of a few ten thousand elements give a wave one tile).  Shapes
    # that stress the carried table: rare deep minima (answers many tiles away), falling runs longer than a tile (every element asks
    # beyond it: more than the kernel keeps per tile), plateaus across tiles (furthest_eq chains that go on beyond the edge), a level
    # per tile (more distinct values at the edges than the table holds).  ansv.hpp:48-65, tie rules ansv_common.hpp:20-22.
    import psac_amd
    rng = np.random.RandomState(31)
    n = (1 << 25) + 777
    NO = (1 << 64) - 1
    band = (10 + rng.geometric(0.35, size=n)).astype(np.uint32)
    deep = np.flatnonzero(rng.rand(n) < 1e-4); band[deep] = rng.randint(0, 10, size=deep.size)
    teeth = (1500 - np.arange(n, dtype=np.uint64) % 1500).astype(np.uint32)
    cuts = np.flatnonzero(rng.rand(n) < 1.0 / 3000)
    plateaus = rng.randint(0, 5, size=cuts.size + 1).astype(np.uint64)[np.searchsorted(cuts, np.arange(n), side="right")]
    levels = ((np.arange(n, dtype=np.uint64) // 1024 * 2654435761 % 40) * 3 + rng.randint(0, 4, size=n).astype(np.uint64)).astype(np.uint32)
    cases = [(band, ((2, 0), (0, 1))), (teeth, ((2, 0), (1, 2))), (plateaus, ((2, 2), (1, 0))), (levels, ((2, 0), (0, 2)))]
    for v, pairs in cases:
        bits = v.dtype.itemsize * 8
        d_in, d_l, d_r = ctx.alloc(n * v.dtype.itemsize), ctx.alloc(n * 8), ctx.alloc(n * 8)
        ctx.h2d(d_in, v)
        want = {}
        for lt, rt in pairs:
            psac_amd.ansv_device(ctx, d_in, n, d_l, d_r, bits, lt, rt, NO)
            L = np.empty(n, np.uint64); R = np.empty(n, np.uint64)
            ctx.d2h(L, d_l); ctx.d2h(R, d_r)
            for side, t, got in ((True, lt, L), (False, rt, R)):
                if (side, t) not in want:
                    want[(side, t)] = O.ansv(v, side, t, NO)
                assert np.array_equal(got, want[(side, t)]), (bits, lt, rt, side)
        for p_ in (d_in, d_l, d_r):
            ctx.free(p_)

