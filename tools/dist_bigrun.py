#!/usr/bin/env python3
"""One rank's share of the 8-GPU config through the distributed step ops: tools/dist_bigrun.py <log2 n> <bits> [P].
P virtual ranks on one GPU; the result is verified by the device checker."""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import torch
import inputs
import psac_amd
from psac_amd import dist as D
from psac_amd.comm import LoopbackWorld
from psac_amd.dist_ops import HipOps

logn = int(sys.argv[1]); bits = int(sys.argv[2]); P = int(sys.argv[3]) if len(sys.argv) > 3 else 1
n = 1 << logn
m = n // P
CH = 1 << 28
parts = []
for o in range(0, n, CH):
    z = inputs.splitmix64_stream(min(CH, n - o), 1 + o)
    parts.append(torch.from_numpy(np.frombuffer(b"ACGT", np.uint8)[(z & np.uint64(3)).astype(np.int64)]).cuda())
text = torch.cat(parts); del parts
ops = [HipOps(bits, 0) for _ in range(P)]
blocks = [text[r * m:(r + 1) * m] for r in range(P)]


def fn(comm, op, blk):
    return (yield from D.construct(comm, op, blk, want_lcp=True))


torch.cuda.synchronize()
t0 = time.time()
res = LoopbackWorld(P).run(fn, [(ops[r], blocks[r]) for r in range(P)])
torch.cuda.synchronize()
dt = time.time() - t0
sa = torch.cat([r["SA"] for r in res]); isa = torch.cat([r["ISA"] for r in res]); lcp = torch.cat([r["LCP"] for r in res])
rounds = res[0]["rounds"]
del res
torch.cuda.synchronize()            # the checker runs on its own stream
ctx = psac_amd.Context(0)
err = psac_amd.check_device(ctx, text.data_ptr(), n, sa.data_ptr(), isa.data_ptr(), lcp.data_ptr(), bits)
print("distributed ops, %d virtual rank(s) x 2^%d / %d, uint%d: %.2f s, rounds %s, peak %.0f GiB, device check errors %s"
      % (P, logn, P, bits, dt, rounds, torch.cuda.max_memory_allocated() / 2**30, err))
