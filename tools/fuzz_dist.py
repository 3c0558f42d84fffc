#!/usr/bin/env python3
"""Randomised parity run of the block-distributed path on ONE GPU: tools/fuzz_dist.py <seconds> [seed].
P virtual ranks (LoopbackWorld) with the HIP step ops; random lengths, alphabets and rank counts; SA, ISA, LCP and
the per-round log are compared with the CPU oracle, and dist_ansv over the LCP with the sequential definition."""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import torch
import oracle_lib as O
from dist_harness import dist as D
from dist_harness.comm import LoopbackWorld
from dist_harness.dist_ops import HipOps

budget = float(sys.argv[1]) if len(sys.argv) > 1 else 60.0
seed = int(sys.argv[2]) if len(sys.argv) > 2 else 1
rng = np.random.RandomState(seed)
t_end = time.time() + budget
runs = 0
while time.time() < t_end:
    P = int(rng.randint(1, 6))
    bits = int(rng.choice([32, 64]))
    sigma = int(rng.choice([1, 2, 4, 20, 127, 256]))
    n = int(rng.choice([rng.randint(200 * P, 5000), rng.randint(5000, 400000), rng.randint(1 << 21, 1 << 22)]))
    lo = int(rng.randint(0, 257 - sigma))
    kind = rng.randint(0, 3)
    if kind == 0:
        text = (lo + rng.randint(0, sigma, size=n)).astype(np.uint8)
    elif kind == 1:
        per = int(rng.randint(1, 3000))
        text = np.tile((lo + rng.randint(0, sigma, size=per)).astype(np.uint8), n // per + 1)[:n].copy()
    else:
        text = (lo + rng.randint(0, sigma, size=n)).astype(np.uint8)
        text[n - int(rng.randint(1, 50)):] = lo
    sizes = D.blk_sizes(n, P); offs = D.prefix(sizes)
    ops = [HipOps(bits, 0) for _ in range(P)]
    blocks = [torch.from_numpy(text[o:o + s].copy()).cuda() for o, s in zip(offs, sizes)]

    def fn(comm, op, blk):
        res = yield from D.construct(comm, op, blk, want_lcp=True)
        L, R = yield from D.dist_ansv(comm, op, res["LCP"], 2, 0)
        return res, L, R
    out = LoopbackWorld(P).run(fn, [(ops[r], blocks[r]) for r in range(P)])
    udt = np.uint32 if bits == 32 else np.uint64
    cat = lambda f: np.concatenate([f(x).cpu().numpy().view(udt) for x in out])
    sa, isa, lcp = cat(lambda x: x[0]["SA"]), cat(lambda x: x[0]["ISA"]), cat(lambda x: x[0]["LCP"])
    L, R = cat(lambda x: x[1]).astype(np.uint64), cat(lambda x: x[2]).astype(np.uint64)
    ref = O.construct(text, bits=bits, fast=False)
    none = (1 << bits) - 1
    ok = np.array_equal(sa, ref["SA"]) and np.array_equal(isa, ref["ISA"]) and np.array_equal(lcp, ref["LCP"])
    ok = ok and [tuple(r) for r in out[0][0]["rounds"]] == [(h, b, e) for h, b, e, _ in ref["trace"]]
    ok = ok and np.array_equal(L, O.ansv(lcp, True, 2, none)) and np.array_equal(R, O.ansv(lcp, False, 0, none))
    for o in ops:
        o.close()
    runs += 1
    if not ok:
        print("MISMATCH: n=%d P=%d uint%d sigma=%d kind=%d" % (n, P, bits, sigma, kind), flush=True)
        np.save("/tmp/fuzz_dist_fail.npy", text)
        sys.exit(1)
print("fuzz_dist: %d runs in %.0f s, all equal to the oracle (seed %d)" % (runs, budget, seed))
