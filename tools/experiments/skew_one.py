#!/usr/bin/env python3
"""tools/experiments/skew_one.py <log2 n>: the repeated-reads text of tools/skewrun.py, uint64, phases only (no check)."""
import os as _os; _os.environ.setdefault("PSACX_ENV_KNOBS", "1")      # PSACX_* variables select the forms of single stages (psac_amd/_lib.py: ENV_KNOBS)
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import psac_amd
n = 1 << int(sys.argv[1])
rng = np.random.RandomState(7)
p = 0.5 ** np.arange(1, 21); p /= p.sum()
_ = (97 + rng.choice(20, size=n, p=p)).astype(np.uint8)          # (same random stream as skewrun.py)
base = np.frombuffer(b"ACGT", np.uint8)[rng.randint(0, 4, size=1 << 16)]
t = np.tile(base, n // base.size + 1)[:n].copy()
mut = rng.randint(0, n, size=n // 200)
t[mut] = np.frombuffer(b"ACGT", np.uint8)[rng.randint(0, 4, size=mut.size)]
ctx = psac_amd.Context(0)
d_text = ctx.alloc(n); ctx.h2d(d_text, t)
d = [ctx.alloc(n * 8) for _ in range(3)]
sa = psac_amd.SuffixArray(index_bits=64, lcp=True, ctx=ctx)
for it in range(2):
    s = sa.construct_device(d_text, n, d[0], d[1], d[2], profile=True)
print("total %.1f ms: keys %.1f scatter %.1f tilehist %.1f rebucket %.1f isa %.1f gather %.1f compact %.1f rmq %.1f" % (
    s.ms_total, s.ms_kmer, s.ms_sort_scatter + s.ms_sort_scatter2 + s.ms_sort_scatter3, s.ms_sort_tilehist, s.ms_rebucket, s.ms_isa_scatter, s.ms_gather, s.ms_compact, s.ms_rmq_build))
