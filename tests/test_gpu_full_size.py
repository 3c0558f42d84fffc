"""Full-size single-GPU configurations, run by the driver's `-m gpu` suite (round 1 only had them in
tools/bigrun.py): BASELINE.json configs[2] (4 GiB random ASCII, uint64), the north-star's headline shape
(4 GiB random DNA, uint64), the largest uint32 input (2^32 - 2 characters), 2 GiB with uint64 indices in the
normal layout, and the /256 twin of configs[4] (period-1024 tandem repeat) on one GPU.

The texts are generated in HBM (psacx_synth_text_dev, the same splitmix64 streams as tests/inputs.py) and
the results are verified where they lie by psacx_check_dev_* -- size-independent properties: SA is a
permutation inverse to ISA, adjacent suffixes are in order, every LCP entry equals a direct character
comparison.  This file sorts before test_gpu_parity.py, so the 283 GiB of the 4 GiB runs are taken while
nothing else holds device memory, and are released at the end of the module.
"""
import ctypes as C
import os

import numpy as np
import pytest

import inputs
import oracle_lib as O

pytestmark = pytest.mark.gpu

KIND = {"dna": 0, "ascii128": 1, "tandem": 2, "mutated": 3}


@pytest.fixture(scope="module")
def ctx():
    import psac_amd
    c = psac_amd.Context(0)
    yield c
    c.close()


def device_text(ctx, n, kind, seed, period=1024):
    d = ctx.alloc(n)
    ctx.check(ctx._lib.psacx_synth_text_dev(ctx.handle, C.c_void_p(d), n, 0, KIND[kind], seed, period))
    return d


def construct_and_check(ctx, n, bits, kind, seed):
    import psac_amd
    w = bits // 8
    d_text = device_text(ctx, n, kind, seed)
    d_sa, d_isa, d_lcp = ctx.alloc(n * w), ctx.alloc(n * w), ctx.alloc(n * w)
    try:
        sa = psac_amd.SuffixArray(index_bits=bits, lcp=True, ctx=ctx)
        s = sa.construct_device(d_text, n, d_sa, d_isa, d_lcp)
        err = psac_amd.check_device(ctx, d_text, n, d_sa, d_isa, d_lcp, bits)
    finally:
        for p in (d_text, d_sa, d_isa, d_lcp):
            ctx.free(p)
    return s, err, sa


def test_synthetic_text_generator_matches_the_host_definition(ctx):
    # the device generator against tests/inputs.py (the definition of SURVEY 8(d)), ragged length and offsets
    n = (1 << 20) + 13
    for kind, ref in (("dna", inputs.dna(n, 5)), ("ascii128", inputs.ascii128(n, 42)),
                      ("tandem", inputs.tandem(n, 1024, inputs.dna(1024, 3))), ("mutated", inputs.mutated(n, 1024, 11))):
        seed = {"dna": 5, "ascii128": 42, "tandem": 3, "mutated": 11}[kind]
        d = device_text(ctx, n, kind, seed)
        got = np.empty(n, np.uint8)
        ctx.d2h(got, d)
        ctx.free(d)
        assert np.array_equal(got, ref), kind
    d = ctx.alloc(1000)
    ctx.check(ctx._lib.psacx_synth_text_dev(ctx.handle, C.c_void_p(d + 3), 990, 777, 0, 9, 1024))   # unaligned start, offset stream
    got = np.empty(1000, np.uint8)
    ctx.d2h(got, d)
    ctx.free(d)
    assert np.array_equal(got[3:993], inputs.dna(777 + 990, 9)[777:])


def test_headline_4gib_dna_uint64(ctx):
    # north-star shape: n = 2^32, uint64, reduced-memory layout, two-stage first round (5 two-word passes)
    s, err, _ = construct_and_check(ctx, 1 << 32, 64, "dna", 1)
    assert err == [0, 0, 0, 0]
    assert s.k == 21 and s.bits_per_char == 3 and s.n_rounds >= 1


def test_config_c3_4gib_ascii_uint64(ctx):
    # BASELINE.json configs[2]: 4 GiB random ASCII (sigma = 128), uint64 indices
    s, err, _ = construct_and_check(ctx, 1 << 32, 64, "ascii128", 42)
    assert err == [0, 0, 0, 0]
    assert s.k == 8 and s.bits_per_char == 8 and s.sigma == 128


def test_largest_uint32_input(ctx):
    # n = 2^32 - 2 (idxsort.hpp:39), one-stage first sort, partition_pairs_kernel above 2^30 records
    s, err, _ = construct_and_check(ctx, (1 << 32) - 2, 32, "dna", 1)
    assert err == [0, 0, 0, 0]
    assert s.k == 10


def test_2gib_dna_uint64_normal_layout(ctx):
    s, err, _ = construct_and_check(ctx, 1 << 31, 64, "dna", 7)
    assert err == [0, 0, 0, 0]


def test_config_c2_bit_exact_vs_divsufsort(ctx):
    # BASELINE.json configs[1]: 256 MiB random DNA, uint32, "bit-exact vs dss": SA against libdivsufsort (the
    # reference's own checker, built into oracle/_ref), ISA by inversion, LCP against Kasai (lcp.hpp:46-77)
    if not O.have_divsufsort():
        pytest.skip("oracle/_ref/libdivsufsort*.so not built")
    import psac_amd
    text = inputs.dna(1 << 28, 1)
    sa = psac_amd.SuffixArray(index_bits=32, lcp=True, ctx=ctx)
    sa.construct(text)
    dSA = O.divsufsort(text, 32)
    assert np.array_equal(sa.local_SA, dSA)
    del dSA
    assert np.array_equal(sa.local_B[sa.local_SA.astype(np.int64)], np.arange(text.size, dtype=np.uint32))
    assert np.array_equal(sa.local_LCP, O.kasai(text, sa.local_SA, sa.local_B))


def _tandem_twin(ctx, monkeypatch, cap):
    # configs[4] / 256: 2^27 characters, period-1024 tandem repeat of DNA(1024, 3), uint64, against libdivsufsort + Kasai (the reference's
    # own checker; the restatement checks the smaller twins of tests/test_gpu_parity.py round by round)
    n = 1 << 27
    text = inputs.tandem(n, 1024, inputs.dna(1024, 3))
    monkeypatch.setenv("PSACX_FORCE_DIET", "1")
    if cap:
        monkeypatch.setenv("PSACX_DIET_CAP", str(cap))
    import psac_amd
    sa = psac_amd.SuffixArray(index_bits=64, lcp=True, ctx=ctx)
    sa.construct(text)
    return text, sa


def test_config_c5_twin_tandem_reduced_memory(ctx, monkeypatch):
    if not O.have_divsufsort():
        pytest.skip("oracle/_ref/libdivsufsort*.so not built")
    monkeypatch.setenv("PSACX_ISA_UPDATE", "levels")          # (the default from 2^31 characters on)
    text, sa = _tandem_twin(ctx, monkeypatch, 0)
    assert np.array_equal(sa.local_B[sa.local_SA.astype(np.int64)], np.arange(text.size, dtype=np.uint64))
    SA, LCP = O.reference_sa_lcp_cached("tandem_1024_3", text, bits=64, isa=sa.local_B)
    assert np.array_equal(sa.local_SA, SA) and np.array_equal(sa.local_LCP, LCP)
    # deep prefix doubling: h = 21 * 2^i until the 1024 phase buckets are resolved
    hs = [r[0] for r in sa.rounds]
    assert hs == [21 << i for i in range(len(hs))] and len(hs) >= 20
    # the same with room for only a quarter of the unresolved suffixes per slab of a refinement round
    text2, sb = _tandem_twin(ctx, monkeypatch, 1 << 25)
    assert np.array_equal(sb.local_SA, SA) and np.array_equal(sb.local_LCP, LCP) and np.array_equal(sb.local_B, sa.local_B)
    # (both runs above take the ISA entries of a round to their places through partition levels, the second collecting them over the
    #  slabs of a round: construct.hpp: IsaLevels) -- the same with one random store per entry
    monkeypatch.setenv("PSACX_ISA_UPDATE", "stores")
    text3, sc = _tandem_twin(ctx, monkeypatch, 1 << 25)
    assert np.array_equal(sc.local_SA, SA) and np.array_equal(sc.local_LCP, LCP) and np.array_equal(sc.local_B, sa.local_B)


def test_host_pointer_path_narrow_transfers(ctx):
    # psacx_construct_u64 on host pointers (SURVEY 8(d) Metric 1: what psac's construct() spans): the results leave the device in the
    # narrowest entries that hold them (engine.hpp: staged_d2h_entries -- SA / ISA of fewer than 2^32 positions in 4 of their 8 bytes, LCP
    # in 1, 2 or 4 according to its largest value) and are widened on the host.  Against the arrays of the device-pointer path copied
    # as they are: random DNA (LCP < 256), one repeat of 300 characters (LCP < 65536), one of 100000 characters (4-byte entries), and
    # 2^30 characters.
    import psac_amd
    def both(text, bits=64):
        n = text.size
        w = bits // 8
        hs = psac_amd.SuffixArray(index_bits=bits, lcp=True, ctx=ctx)
        hs.construct(text)
        d_text = ctx.alloc(n); ctx.h2d(d_text, text)
        d = [ctx.alloc(n * w) for _ in range(3)]
        try:
            ds = psac_amd.SuffixArray(index_bits=bits, lcp=True, ctx=ctx)
            ds.construct_device(d_text, n, d[0], d[1], d[2])
            tmp = np.empty(n, np.uint64 if bits == 64 else np.uint32)
            for arr, p in ((hs.local_SA, d[0]), (hs.local_B, d[1]), (hs.local_LCP, d[2])):
                ctx.d2h(tmp, p)
                assert np.array_equal(arr, tmp)
        finally:
            for p in [d_text] + d:
                ctx.free(p)
        return hs
    # (round 6: when the first round leaves no bucket unresolved, SA and LCP start on their way out under its SA -> ISA inversion,
    #  construct.hpp: EarlyOut -- the random texts below at 64 bits; with repeats, or a few equal 2k-mers at 32 bits, they wait)
    n = (1 << 27) + 12345
    t = inputs.dna(n, 21)
    hs = both(t)
    assert int(hs.local_LCP.max()) < 256
    os.environ["PSACX_NO_EARLY_OUT"] = "1"          # (the test suite's debug shim: psac_amd/_lib.py ENV_KNOBS)
    try:
        both(t)
    finally:
        del os.environ["PSACX_NO_EARLY_OUT"]
    both(t, bits=32)
    both(inputs.ascii128(n, 3), bits=32)
    t2 = t.copy(); t2[90000000:90000300] = t2[1000:1300]
    hs = both(t2)
    assert 300 <= int(hs.local_LCP.max()) < 65536
    t3 = t.copy(); t3[90000000:90100000] = t3[5000:105000]
    hs = both(t3)
    assert int(hs.local_LCP.max()) >= 100000
    del hs, t2, t3
    both(inputs.dna(1 << 30, 4))
