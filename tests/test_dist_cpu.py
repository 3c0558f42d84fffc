"""The distributed choreography (tests/dist_harness/dist.py) on the CPU: virtual ranks in one process
(LoopbackWorld) for many rank counts and inputs, and two real processes over gloo."""
import os
import sys

import numpy as np
import pytest
import torch

import inputs
import oracle_lib as O
from numpy_ops import NumpyOps
from dist_harness import dist as D
from dist_harness.comm import LoopbackWorld

HERE = os.path.dirname(os.path.abspath(__file__))


def split_blocks(text, P):
    sizes = D.blk_sizes(text.size, P)
    offs = D.prefix(sizes)
    return [torch.from_numpy(text[o:o + s].copy()) for o, s in zip(offs, sizes)]


def run_loopback(text, P, bits, k=0, want_lcp=True):
    blocks = split_blocks(text, P)

    def fn(comm, blk):
        return (yield from D.construct(comm, NumpyOps(bits), blk, want_lcp=want_lcp, k_req=k))
    res = LoopbackWorld(P).run(fn, [(b,) for b in blocks])
    ops = NumpyOps(bits)
    cat = lambda key: np.concatenate([ops.u(r[key]) for r in res])
    return cat("SA"), cat("ISA"), (cat("LCP") if want_lcp else None), res[0]


def check_against_oracle(text, P, bits, k=0):
    sa, isa, lcp, info = run_loopback(text, P, bits, k=k)
    ref = O.construct(text, bits=bits, k=k if P == 1 else 0) if k == 0 else None
    if ref is None:
        ref_sa = O.naive_sa(text, bits)
        assert np.array_equal(sa, ref_sa)
        assert O.check_sa(text, sa, isa) == 0
        assert np.array_equal(O.kasai(text, sa, isa), lcp)
    else:
        assert np.array_equal(sa, ref["SA"]) and np.array_equal(isa, ref["ISA"]) and np.array_equal(lcp, ref["LCP"])
    return info


@pytest.mark.parametrize("P", [1, 2, 3, 4, 7])
def test_loopback_random_dna(P):
    text = O.rand_dna(5003, 7)
    for bits in (32, 64):
        check_against_oracle(text, P, bits)


@pytest.mark.parametrize("P", [2, 3, 5])
def test_loopback_repetitive_and_small_k(P):
    # deep rounds (tandem repeat), forced bucket refinement (k = 3) and heavy ties
    unit = O.rand_dna(64, 3)
    check_against_oracle(inputs.tandem(4000, 64, unit), P, 32)
    check_against_oracle(O.rand_dna(3001, 23), P, 64, k=3)
    check_against_oracle(np.frombuffer(b"A" * 900, np.uint8).copy(), P, 32)
    check_against_oracle(inputs.cyclic(1500, "abc"), P, 64)
    check_against_oracle(inputs.ascii128(2500, 9), P, 64)


@pytest.mark.parametrize("P", [1, 2, 3])
def test_loopback_short_suffix_ties(P):
    # texts whose tail repeats the smallest character: the suffixes shorter than 2k tie with longer ones
    # on the zero padding of the packed window and must still sort first (they are moved to rank 0)
    a = np.frombuffer(b"A", np.uint8)
    for text in (np.concatenate([O.rand_dna(3000, 5), np.repeat(a, 40)]), np.repeat(a, 900),
                 np.concatenate([np.repeat(a, 500), O.rand_dna(700, 2), np.repeat(a, 25)]),
                 np.concatenate([inputs.ascii128(2000, 3), np.zeros(33, np.uint8)]),
                 inputs.cyclic(1500, "AAC")):
        for bits in (32, 64):
            check_against_oracle(text, P, bits)


def _ansv_loopback(vals, P, bits, lt, rt, make_ops, to_tensor, to_numpy):
    sizes = D.blk_sizes(vals.size, P)
    offs = D.prefix(sizes)
    blocks = [to_tensor(vals[o:o + s]) for o, s in zip(offs, sizes)]

    def fn(comm, ops, blk):
        return (yield from D.dist_ansv(comm, ops, blk, lt, rt))
    res = LoopbackWorld(P).run(fn, [(make_ops(), b) for b in blocks])
    return (np.concatenate([to_numpy(x[0]) for x in res]).astype(np.uint64),
            np.concatenate([to_numpy(x[1]) for x in res]).astype(np.uint64))


@pytest.mark.parametrize("P", [1, 2, 3, 5])
def test_loopback_distributed_ansv(P):
    # ansv<T, left_type, right_type, global_indexing> (ansv.hpp:2042-2051) over blocks: all nine type pairs
    # (ansv_common.hpp:20-22) against the sequential definition, on inputs with many ties and few ties
    rng = np.random.RandomState(P)
    for trial in range(6):
        n = int(rng.randint(P, 400))
        vals = rng.randint(0, int(rng.choice([2, 5, 1000])), size=n).astype(np.uint64)
        for bits in (32, 64):
            ops = NumpyOps(bits)
            v = vals.astype(np.uint32 if bits == 32 else np.uint64)
            none = (1 << bits) - 1
            for lt in (0, 1, 2):
                for rt in (0, 1, 2):
                    L, R = _ansv_loopback(vals, P, bits, lt, rt, lambda: NumpyOps(bits), ops.t, ops.u)
                    assert np.array_equal(L, O.ansv(v, True, lt, none)), (n, bits, lt)
                    assert np.array_equal(R, O.ansv(v, False, rt, none)), (n, bits, rt)
    # the LCP array of a text: the input psac feeds it (suffix_tree.hpp:62: left furthest_eq, right nearest_sm)
    text = O.rand_dna(3000, 4)
    r = O.construct(text, bits=64)
    ops = NumpyOps(64)
    L, R = _ansv_loopback(r["LCP"], P, 64, 2, 0, lambda: NumpyOps(64), ops.t, ops.u)
    assert np.array_equal(L, O.ansv(r["LCP"], True, 2, (1 << 64) - 1)) and np.array_equal(R, O.ansv(r["LCP"], False, 0, (1 << 64) - 1))


def test_loopback_round_log_matches_oracle():
    text = inputs.tandem(6000, 128, O.rand_dna(128, 5))
    info = check_against_oracle(text, 3, 32)
    ref = O.construct(text, bits=32, fast=False)     # full doubling prints every round
    assert [(h, b, e) for (h, b, e) in info["rounds"]] == [(h, b, e) for (h, b, e, _) in ref["trace"]]


def test_shift_by_h_saturates_instead_of_wrapping():
    # 32-bit indices, n > 2^31: SA + h of a deep round (suffix_array.hpp:978) must become "past the end" (>= n),
    # never a small in-range position (the wrap would fetch ISA[(SA + h) mod 2^32] + 1 instead of key 0)
    ops = NumpyOps(32)
    n = 0xFFFFFF00
    sa = ops.t(np.array([5, 0x80000000, 0xFFFFFE00, 0xFFFFFEFF], np.uint32))
    q = ops.u(ops.add_scalar(sa, 0x200, n))
    assert q.tolist() == [0x205, 0x80000200, n, n]
    assert ops.u(ops.finish_b2(ops.t(np.array([7, 8, 9, 10], np.uint32)), ops.t(q), n)).tolist() == [8, 9, 0, 0]


def test_block_distribution_is_enforced():
    text = O.rand_dna(1000, 1)

    def fn(comm, blk):
        return (yield from D.construct(comm, NumpyOps(32), blk))
    blocks = [torch.from_numpy(text[:600].copy()), torch.from_numpy(text[600:].copy())]
    with pytest.raises(RuntimeError):
        LoopbackWorld(2).run(fn, [(b,) for b in blocks])


def _gloo_worker(rank, world, port, text, bits, out):
    import torch.distributed as dist
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    sys.path.insert(0, HERE)
    from numpy_ops import NumpyOps as Ops
    from dist_harness import dist as DD
    from dist_harness.comm import TorchComm
    blk = split_blocks(text, world)[rank]
    gen = DD.construct(TorchComm(), Ops(bits), blk)
    try:
        while True:
            next(gen)
    except StopIteration as e:
        res = e.value
    ops = Ops(bits)
    out[rank] = (ops.u(res["SA"]).copy(), ops.u(res["ISA"]).copy(), ops.u(res["LCP"]).copy())
    dist.destroy_process_group()


def test_gloo_two_processes():
    import torch.multiprocessing as mp
    text = O.rand_dna(4001, 11)
    mgr = mp.Manager()
    out = mgr.dict()
    port = 29500 + (os.getpid() % 2000)
    mp.spawn(_gloo_worker, args=(2, port, text, 32, out), nprocs=2, join=True)
    sa = np.concatenate([out[0][0], out[1][0]])
    isa = np.concatenate([out[0][1], out[1][1]])
    lcp = np.concatenate([out[0][2], out[1][2]])
    ref = O.construct(text, bits=32)
    assert np.array_equal(sa, ref["SA"]) and np.array_equal(isa, ref["ISA"]) and np.array_equal(lcp, ref["LCP"])


def _gloo_ansv_worker(rank, world, port, vals, bits, out):
    import torch.distributed as dist
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    sys.path.insert(0, HERE)
    from numpy_ops import NumpyOps as Ops
    from dist_harness import dist as DD
    from dist_harness.comm import TorchComm
    ops = Ops(bits)
    sizes = DD.blk_sizes(vals.size, world)
    offs = DD.prefix(sizes)
    blk = ops.t(vals[offs[rank]:offs[rank] + sizes[rank]])
    L, R = DD.run(DD.dist_ansv(TorchComm(), ops, blk, 2, 0))        # the pair psac's suffix tree uses
    out[rank] = (ops.u(L).copy(), ops.u(R).copy())
    dist.destroy_process_group()


def test_gloo_two_processes_ansv():
    import torch.multiprocessing as mp
    text = O.rand_dna(3001, 2)
    lcp = O.construct(text, bits=64)["LCP"]
    mgr = mp.Manager()
    out = mgr.dict()
    port = 31500 + (os.getpid() % 2000)
    mp.spawn(_gloo_ansv_worker, args=(2, port, lcp, 64, out), nprocs=2, join=True)
    L = np.concatenate([out[0][0], out[1][0]]).astype(np.uint64)
    R = np.concatenate([out[0][1], out[1][1]]).astype(np.uint64)
    none = (1 << 64) - 1
    assert np.array_equal(L, O.ansv(lcp, True, 2, none)) and np.array_equal(R, O.ansv(lcp, False, 0, none))
