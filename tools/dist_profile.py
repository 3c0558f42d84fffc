#!/usr/bin/env python3
"""Where the distributed path spends its time: tools/dist_profile.py [log2 n per rank] [P] [bits].
Runs dist.construct for P virtual ranks on ONE GPU (LoopbackWorld), synchronising around every local
op and every collective, and prints the time per op name summed over ranks.  (Collectives here are
device copies inside one process; the table is about the local kernels and the host-side glue.)"""
import collections
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import torch
import inputs
from dist_harness import dist as D
from dist_harness.comm import LoopbackWorld
from dist_harness.dist_ops import HipOps

logn = int(sys.argv[1]) if len(sys.argv) > 1 else 26
P = int(sys.argv[2]) if len(sys.argv) > 2 else 2
bits = int(sys.argv[3]) if len(sys.argv) > 3 else 32
m = 1 << logn
acc = collections.defaultdict(float)
calls = collections.defaultdict(int)


def timed(name, fn):
    def w(*a, **k):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        r = fn(*a, **k)
        torch.cuda.synchronize()
        acc[name] += time.perf_counter() - t0
        calls[name] += 1
        return r
    return w


ops = [HipOps(bits, 0) for _ in range(P)]
for o in ops:
    for name in dir(o):
        if name.startswith("_") or name in ("close", "profile", "stats", "index_bits"):
            continue
        f = getattr(o, name)
        if callable(f):
            setattr(o, name, timed(name, f))
blocks = [torch.from_numpy(inputs.dna(m, 1 + r)).cuda() for r in range(P)]


def fn(comm, op, blk):
    return (yield from D.construct(comm, op, blk, want_lcp=True))


for it in range(2):
    acc.clear(); calls.clear()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    res = LoopbackWorld(P).run(fn, [(ops[r], blocks[r]) for r in range(P)])
    torch.cuda.synchronize()
    total = time.perf_counter() - t0
print("n per rank 2^%d, P=%d, uint%d: %.1f ms wall for all ranks (with per-op syncs), rounds %s" % (logn, P, bits, total * 1e3, res[0]["rounds"]))
tsum = 0.0
for name, t in sorted(acc.items(), key=lambda kv: -kv[1]):
    print("  %-18s %8.2f ms  %4d calls" % (name, t * 1e3, calls[name]))
    tsum += t
print("  %-18s %8.2f ms" % ("(outside ops)", (total - tsum) * 1e3))
