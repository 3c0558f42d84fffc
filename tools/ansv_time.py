#!/usr/bin/env python3
"""ANSV over an LCP array resident in HBM: tools/ansv_time.py <log2 n> <bits>.  Constructs SA+LCP of random DNA,
then times psacx_ansv_dev_* (left furthest_eq, right nearest_sm: the pair psac -t uses)."""
import os as _os; _os.environ.setdefault("PSACX_ENV_KNOBS", "1")      # PSACX_* variables select the forms of single stages (psac_amd/_lib.py: ENV_KNOBS)
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import inputs
import psac_amd

logn = int(sys.argv[1]); bits = int(sys.argv[2])
n = 1 << logn; w = bits // 8
ctx = psac_amd.Context(0)
text = inputs.dna(n, 1)
d_text = ctx.alloc(n); ctx.h2d(d_text, text)
d_sa, d_isa, d_lcp = ctx.alloc(n * w), ctx.alloc(n * w), ctx.alloc(n * w)
d_l, d_r = ctx.alloc(n * 8), ctx.alloc(n * 8)
sa = psac_amd.SuffixArray(index_bits=bits, lcp=True, ctx=ctx)
sa.construct_device(d_text, n, d_sa, d_isa, d_lcp)
names = ("nearest_sm", "nearest_eq", "furthest_eq")
mode = sys.argv[3] if len(sys.argv) > 3 else ""
pairs = ((0, 0),) if mode == "one" else ((2, 0),) if mode == "t" else ((2, 0), (0, 0), (1, 1), (2, 2))          # one: nearest_sm pair; t: the pair psac -t uses
for lt, rt in pairs:
    for it in range(3):
        t0 = time.perf_counter()
        psac_amd.ansv_device(ctx, d_lcp, n, d_l, d_r, bits, lt, rt, (1 << 64) - 1)
        dt = time.perf_counter() - t0
    # algorithmic bytes: the input once, both uint64 results once
    print("ANSV(%s, %s) over the LCP of 2^%d random DNA characters, uint%d, HBM-resident: %.2f ms = %.1f G elements/s = %.0f GB/s algorithmic"
          % (names[lt], names[rt], logn, bits, dt * 1e3, n / dt / 1e9, n * (w + 16) / dt / 1e9))
