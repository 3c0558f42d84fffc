#!/usr/bin/env python3
"""Generalized suffix array at scale: tools/gsa_time.py <log2 total characters> <read length> <bits>.
Random DNA reads of equal length; SA+ISA+LCP through psacx_construct_gsa_*; spot-checks order and LCP."""
import os as _os; _os.environ.setdefault("PSACX_ENV_KNOBS", "1")      # PSACX_* variables select the forms of single stages (psac_amd/_lib.py: ENV_KNOBS)
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import inputs
import psac_amd

logn = int(sys.argv[1]); rl = int(sys.argv[2]); bits = int(sys.argv[3])
n = (1 << logn) // rl * rl
text = inputs.dna(n, 3)
off = np.arange(0, n + 1, rl, dtype=np.uint64)
ctx = psac_amd.Context(0)
sa = psac_amd.SuffixArray(index_bits=bits, lcp=True, ctx=ctx)
lib = ctx._lib
SA = np.empty(n, sa.dtype); ISA = np.empty(n, sa.dtype); LCP = np.empty(n, sa.dtype)
fn = getattr(lib, "psacx_construct_gsa_u%d" % bits)
for it in range(2):
    t0 = time.perf_counter()
    ctx.check(fn(ctx.handle, text.ctypes.data, n, off.ctypes.data, off.size - 1, 0, 1 | 4, SA.ctypes.data, ISA.ctypes.data, LCP.ctypes.data))
    dt = time.perf_counter() - t0
s = ctx.stats()
rng = np.random.RandomState(1)
bad = 0
for i in rng.randint(1, n, size=2000):
    a, b = int(SA[i - 1]), int(SA[i])
    ea, eb = (a // rl + 1) * rl, (b // rl + 1) * rl
    x, y = bytes(text[a:ea]), bytes(text[b:eb])
    c = 0
    while c < len(x) and c < len(y) and x[c] == y[c]:
        c += 1
    bad += not ((x < y or (x == y and a < b)) and int(LCP[i]) == c)
print("GSA of %d reads x %d (n = %d), uint%d: %.1f ms host call (device %.1f ms), %d rounds, sample check errors %d"
      % (n // rl, rl, n, bits, dt * 1e3, s.ms_total, s.n_rounds, bad))
