#!/usr/bin/env python3
"""Times the stand-alone rank-pair sort (psacx_pair_sort_dev) on round-1-like keys:
(B1,B2) = (10-mer at i, 10-mer at i+10) of random DNA, 3 bits per character.
"""
import os as _os; _os.environ.setdefault("PSACX_ENV_KNOBS", "1")      # PSACX_* variables select the forms of single stages (psac_amd/_lib.py: ENV_KNOBS)
import ctypes as C
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import inputs
import psac_amd

logn = int(sys.argv[1]) if len(sys.argv) > 1 else 28
bits = int(sys.argv[2]) if len(sys.argv) > 2 else 32
reps = int(sys.argv[3]) if len(sys.argv) > 3 else 3
n = 1 << logn
dt = np.uint32 if bits == 32 else np.uint64
k, l = (10, 3) if bits == 32 else (21, 3)
code = (inputs.splitmix64_stream(n + 2 * k, 1) & np.uint64(3)).astype(np.uint64) + np.uint64(1)
b = np.zeros(n + k, np.uint64)
for j in range(k):
    b = (b << np.uint64(l)) | code[j:j + n + k]
b1 = b[:n].astype(dt); b2 = b[k:k + n].astype(dt)
ctx = psac_amd.Context(0)
w = bits // 8
d1 = ctx.alloc(n * w); d2 = ctx.alloc(n * w); di = ctx.alloc(n * w)
fn = getattr(ctx._lib, "psacx_pair_sort_dev_u%d" % bits)
best = None
for r in range(reps):
    ctx.h2d(d1, b1); ctx.h2d(d2, b2)
    ctx.check(fn(ctx.handle, C.c_void_p(d1), C.c_void_p(d2), C.c_void_p(di), n, k * l))
    s = ctx.stats()
    q = 1 if s.scatter_bytes[1] else 0
    ms = s.ms_sort_scatter3 if q else s.ms_sort_scatter
    per = ms / max(s.scatter_launches[q], 1)
    gbs = s.scatter_bytes[q] / (ms * 1e-3) / 1e9
    print("cfg=%s n=2^%d u%d: hist %.3f ms, tile-hist %.3f ms, scatter %.3f ms over %d passes (%.3f ms/pass) -> %.0f GB/s algorithmic (%.1f%% of 8 TB/s)"
          % ("def", logn, bits, s.ms_sort_hist, s.ms_sort_tilehist, ms, s.scatter_launches[q], per, gbs, gbs / 80.0))
if logn <= 24:
    o1 = np.empty(n, dt); o2 = np.empty(n, dt); oi = np.empty(n, dt)
    ctx.d2h(o1, d1); ctx.d2h(o2, d2); ctx.d2h(oi, di)
    order = np.lexsort((b2, b1))
    assert np.array_equal(oi, order.astype(dt)) and np.array_equal(o1, b1[order]) and np.array_equal(o2, b2[order])
    print("verified vs numpy lexsort")
