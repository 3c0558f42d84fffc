"""The exchanges of the multi-GPU engine's first round and of its sample sort, carried out by REAL processes over gloo exactly as the
shipped planning code (psac_amd/csrc/multi_plan.hpp, the functions multi.hpp calls before it issues ncclSend / ncclRecv) says: every rank
asks the plan for its sends and receives, moves numpy arrays with torch.distributed point-to-point calls, and must end with its block of the
globally sorted records.  tests/cpp/plan_capi.cpp (g++, no HIP) exposes the plan functions; the device kernels are replaced by numpy
(a stable partition by top digit, a stable sort inside the buckets)."""
import ctypes as C
import os
import subprocess
import sys

import numpy as np
import pytest
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
_SO = None


def plan_lib():
    global _SO
    if _SO is None:
        so = os.path.join(HERE, "cpp", "libplan_capi.so")
        src = os.path.join(HERE, "cpp", "plan_capi.cpp")
        hdr = os.path.join(os.path.dirname(HERE), "psac_amd", "csrc", "multi_plan.hpp")
        if not os.path.exists(so) or os.path.getmtime(so) < max(os.path.getmtime(src), os.path.getmtime(hdr)):
            subprocess.check_call(["g++", "-std=c++14", "-O1", "-Wall", "-shared", "-fPIC", "-o", so, src])
        _SO = C.CDLL(so)
        _SO.plan_bucket_start.restype = C.c_uint64
    return _SO


def _u64(a):
    return np.ascontiguousarray(a, dtype=np.uint64)


def _p(a):
    return a.ctypes.data_as(C.c_void_p)


def _exchange(dist, sends, recvs, src_arr, dst_arr):
    """sends / recvs: lists of (peer, offset, count).  Messages between a pair of ranks match in order, as ncclSend / ncclRecv do."""
    me = dist.get_rank()
    reqs, bufs = [], []
    for peer, off, cnt in recvs:
        if peer == me:
            continue
        t = torch.empty(int(cnt), dtype=torch.int64)
        reqs.append(dist.irecv(t, src=int(peer)))
        bufs.append((t, int(off), int(cnt)))
    for peer, off, cnt in sends:
        if peer == me:
            continue
        reqs.append(dist.isend(torch.from_numpy(src_arr[int(off):int(off) + int(cnt)].astype(np.int64)), dst=int(peer)))
    # a rank's messages to itself: matched in order
    mine_s = [(o, c) for p_, o, c in sends if p_ == me]
    mine_r = [(o, c) for p_, o, c in recvs if p_ == me]
    assert [c for _, c in mine_s] == [c for _, c in mine_r]
    staged = [src_arr[int(o):int(o) + int(c)].copy() for o, c in mine_s]
    for r_ in reqs:
        r_.wait()
    for t, off, cnt in bufs:
        dst_arr[off:off + cnt] = t.numpy().astype(np.uint64)
    for (o, c), piece in zip(mine_r, staged):
        dst_arr[int(o):int(o) + int(c)] = piece


def _first_round_worker(rank, world, port, kind, out):
    import torch.distributed as dist
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    L = plan_lib()
    P, QR = world, 3
    n = 60000 * P + (1 if P > 2 else 0)
    sizes = np.zeros(P, np.uint64)
    L.plan_blk(C.c_uint64(n), C.c_uint(P), _p(sizes))
    offs = np.concatenate([[0], np.cumsum(sizes)]).astype(np.uint64)
    rng = np.random.RandomState(17)                      # (every rank draws the whole key array and keeps its block)
    if kind == "uniform":
        digit = rng.randint(0, 256, n)
    else:                                                # skewed digit counts, the largest bucket about 1 % of the text
        w = 1.0 + 3.0 * (np.arange(256) % 5 == 0) + (np.arange(256) // 64) * 0.5
        digit = rng.choice(256, size=n, p=w / w.sum())
    low = rng.randint(0, 1 << 20, n)
    key = (digit.astype(np.uint64) << np.uint64(40)) | (low.astype(np.uint64) << np.uint64(20))      # ties on (digit, low) keep suffix order
    lo, hi = int(offs[rank]), int(offs[rank + 1])
    rec = key[lo:hi] + np.arange(lo, hi, dtype=np.uint64)                                            # the suffix in the low 20 bits (n < 2^20)
    mine_digit = digit[lo:hi]
    # the sender's partition by top digit (key_scatter1w_kernel: stable)
    order = np.argsort(mine_digit, kind="stable")
    grp = rec[order]
    counts = np.bincount(mine_digit, minlength=256).astype(np.uint64)
    gathered = [torch.zeros(256, dtype=torch.int64) for _ in range(P)]
    dist.all_gather(gathered, torch.from_numpy(counts.astype(np.int64)))
    table = _u64(np.stack([g.numpy() for g in gathered]).astype(np.uint64))
    shorts = np.zeros(256, np.uint64)
    cut = np.zeros(P + 1, np.int32); Gs = np.zeros(P, np.uint64); cs = np.zeros(P, np.uint64); Hs = np.zeros(P, np.uint64); rooms = np.zeros(P, np.uint64)
    inplace = C.c_int(0)
    rc = L.plan_deal(_p(table), P, _p(shorts), _p(sizes), 0, QR, _p(cut), _p(Gs), _p(cs), _p(Hs), _p(rooms), C.byref(inplace))
    assert rc == 0 and inplace.value == 1
    A = np.zeros(int(rooms[rank]), np.uint64)
    so, ro, cn = (np.zeros(256, np.uint64) for _ in range(3))

    def pieces(r, d, q):
        k = L.plan_pieces(_p(table), P, _p(shorts), _p(sizes), 0, QR, r, d, q, _p(so), _p(ro), _p(cn))
        return [(int(so[i]), int(ro[i]), int(cn[i])) for i in range(k)]
    for q in range(QR):
        sends = [(d, s_, c_) for d in range(P) for s_, _, c_ in pieces(rank, d, q)]
        recvs = [(r, r_, c_) for r in range(P) for _, r_, c_ in pieces(r, rank, q)]
        _exchange(dist, sends, recvs, grp, A)
    # the LSD passes inside the buckets (stable): every bucket of the share sorted by the rest of the key
    share = A[int(Hs[rank]):int(Hs[rank]) + int(cs[rank])]
    assert np.all(share != 0) or (rank == 0 and np.count_nonzero(share == 0) <= 1)
    b_of = (share >> np.uint64(40)).astype(np.int64)
    assert np.all(np.diff(b_of) >= 0) and b_of.min() >= cut[rank] and b_of.max() < cut[rank + 1]
    share_sorted = share[np.argsort(share >> np.uint64(20), kind="stable")]
    A[int(Hs[rank]):int(Hs[rank]) + int(cs[rank])] = share_sorted
    # the re-balance in place
    TP = _u64(offs)
    sends = np.zeros(3 * P, np.int64); recvs = np.zeros(3 * P, np.int64); ns = C.c_int(0); nr = C.c_int(0)
    assert L.plan_in_place(rank, P, _p(Gs), _p(cs), _p(TP), C.c_uint64(int(Hs[rank])), _p(sends), C.byref(ns), _p(recvs), C.byref(nr)) == 0
    snap = A.copy()
    _exchange(dist, [tuple(sends[3 * i:3 * i + 3]) for i in range(ns.value)], [tuple(recvs[3 * i:3 * i + 3]) for i in range(nr.value)], snap, A)
    # (the rank's own records move to the front of the arrays: multi.hpp rewinds the array by the headroom instead)
    own_lo = max(int(Gs[rank]), lo) - int(Gs[rank]) + int(Hs[rank])
    own_cnt = min(int(Gs[rank]) + int(cs[rank]), hi) - max(int(Gs[rank]), lo)
    A[max(int(Gs[rank]), lo) - lo:max(int(Gs[rank]), lo) - lo + own_cnt] = snap[own_lo:own_lo + own_cnt]
    allrec = key + np.arange(n, dtype=np.uint64)
    want = allrec[np.argsort(allrec >> np.uint64(20), kind="stable")][lo:hi]
    out[rank] = bool(np.array_equal(A[:hi - lo], want))
    dist.destroy_process_group()


@pytest.mark.parametrize("world,kind", [(2, "uniform"), (3, "skewed")])
def test_first_round_shuffle_and_rebalance_over_gloo(world, kind):
    import torch.multiprocessing as mp
    plan_lib()
    mgr = mp.Manager()
    out = mgr.dict()
    port = 33500 + (os.getpid() % 2000) + world
    mp.spawn(_first_round_worker, args=(world, port, kind, out), nprocs=world, join=True)
    assert [out[r] for r in range(world)] == [True] * world


def _sample_sort_worker(rank, world, port, out):
    import torch.distributed as dist
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    L = plan_lib()
    P = world
    n = 20000 * P + 1
    sizes = np.zeros(P, np.uint64)
    L.plan_blk(C.c_uint64(n), C.c_uint(P), _p(sizes))
    offs = np.concatenate([[0], np.cumsum(sizes)]).astype(np.uint64)
    rng = np.random.RandomState(23)
    k1 = rng.randint(0, 50, n).astype(np.uint64)          # many equal keys: ties are divided by (rank, index)
    k2 = rng.randint(0, 3, n).astype(np.uint64)
    lo, hi = int(offs[rank]), int(offs[rank + 1])
    a1, a2 = k1[lo:hi].copy(), k2[lo:hi].copy()
    payload = np.arange(lo, hi, dtype=np.uint64)
    SAMPLES = 64
    pos = np.zeros(SAMPLES, np.uint64)
    k = L.plan_sample_positions(C.c_uint64(hi - lo), rank, C.c_uint64(5), SAMPLES, _p(pos))
    smp = np.zeros((SAMPLES, 4), np.uint64)
    for i in range(k):
        smp[i] = (a1[int(pos[i])], a2[int(pos[i])], rank, pos[i])
    cnts = [torch.zeros(1, dtype=torch.int64) for _ in range(P)]
    dist.all_gather(cnts, torch.tensor([k], dtype=torch.int64))
    allsmp = [torch.zeros((SAMPLES, 4), dtype=torch.int64) for _ in range(P)]
    dist.all_gather(allsmp, torch.from_numpy(smp.astype(np.int64)))
    flat = _u64(np.concatenate([allsmp[r].numpy()[:int(cnts[r][0])] for r in range(P)]).astype(np.uint64))
    spl = np.zeros((P, 4), np.uint64)
    ns = L.plan_splitters(_p(flat), flat.shape[0], P, _p(spl))
    dest = np.zeros(hi - lo, np.uint32)
    L.plan_destinations(_p(spl), ns, _p(a1), _p(a2), C.c_uint64(hi - lo), C.c_uint64(rank), _p(dest))
    order = np.argsort(dest, kind="stable")                # the stable partition pass (op_split_by)
    bounds = np.concatenate([[0], np.cumsum(np.bincount(dest, minlength=P))]).astype(np.int64)
    packed = (a1 << np.uint64(40)) | (a2 << np.uint64(32)) | payload            # one array on the wire here (three in the engine)
    grp = packed[order]
    # all-to-all: counts, then the pieces
    c_all = [torch.zeros(P, dtype=torch.int64) for _ in range(P)]
    dist.all_gather(c_all, torch.from_numpy(np.diff(bounds).astype(np.int64)))
    rc = [int(c_all[r][rank]) for r in range(P)]
    got = np.zeros(sum(rc), np.uint64)
    roff = np.concatenate([[0], np.cumsum(rc)]).astype(np.int64)
    _exchange(dist, [(d, int(bounds[d]), int(bounds[d + 1] - bounds[d])) for d in range(P)], [(r, int(roff[r]), rc[r]) for r in range(P)], grp, got)
    got = got[np.argsort(got >> np.uint64(32), kind="stable")]               # the local sort by (k1, k2)
    # exact re-balance to the block sizes
    tot = [torch.zeros(1, dtype=torch.int64) for _ in range(P)]
    dist.all_gather(tot, torch.tensor([got.size], dtype=torch.int64))
    G = np.concatenate([[0], np.cumsum([int(t[0]) for t in tot])]).astype(np.uint64)
    TP = _u64(offs)
    b = np.zeros(P + 1, np.uint64)
    L.plan_rebalance_bounds(C.c_uint64(int(G[rank])), C.c_uint64(got.size), _p(TP), P, _p(b))
    sends = [(d, int(b[d]), int(b[d + 1] - b[d])) for d in range(P)]
    ball = [torch.zeros(P + 1, dtype=torch.int64) for _ in range(P)]
    dist.all_gather(ball, torch.from_numpy(b.astype(np.int64)))
    rc2 = [int(ball[r][rank + 1] - ball[r][rank]) for r in range(P)]
    fin = np.zeros(sum(rc2), np.uint64)
    ro2 = np.concatenate([[0], np.cumsum(rc2)]).astype(np.int64)
    _exchange(dist, sends, [(r, int(ro2[r]), rc2[r]) for r in range(P)], got, fin)
    allp = (k1 << np.uint64(40)) | (k2 << np.uint64(32)) | np.arange(n, dtype=np.uint64)
    ref = np.sort(allp >> np.uint64(32))[lo:hi]                                # the keys of this block of the global order
    out[rank] = bool(fin.size == hi - lo and np.array_equal(fin >> np.uint64(32), ref))
    dist.destroy_process_group()


def test_sample_sort_and_exact_rebalance_over_gloo():
    import torch.multiprocessing as mp
    plan_lib()
    mgr = mp.Manager()
    out = mgr.dict()
    port = 35500 + (os.getpid() % 2000)
    mp.spawn(_sample_sort_worker, args=(2, port, out), nprocs=2, join=True)
    assert [out[r] for r in range(2)] == [True, True]
