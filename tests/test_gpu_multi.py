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
        rSA, rLCP = O.construct_all_cores(text, bits=64)
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
