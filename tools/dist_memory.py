#!/usr/bin/env python3
"""Peak device memory of the block-distributed path: tools/dist_memory.py <log2 n per rank> <bits> <P> (virtual ranks
on one GPU)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch, inputs
from dist_harness import dist as D
from dist_harness.comm import LoopbackWorld
from dist_harness.dist_ops import HipOps
logn = int(sys.argv[1]); bits = int(sys.argv[2]); P = int(sys.argv[3])
m = 1 << logn
ops = [HipOps(bits, 0) for _ in range(P)]
blocks = [torch.from_numpy(inputs.dna(m, 1 + r)).cuda() for r in range(P)]
def fn(comm, op, blk):
    return (yield from D.construct(comm, op, blk, want_lcp=True))
torch.cuda.reset_peak_memory_stats()
res = LoopbackWorld(P).run(fn, [(ops[r], blocks[r]) for r in range(P)])
torch.cuda.synchronize()
peak = torch.cuda.max_memory_allocated() / 2**30
slab = sum(o.stats().workspace_bytes for o in ops) / 2**30
print("P=%d x 2^%d uint%d: torch peak %.1f GiB (%.1f bytes per character per rank incl. results), op workspaces %.1f GiB"
      % (P, logn, bits, peak, peak * 2**30 / (P * m), slab))
