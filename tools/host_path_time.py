#!/usr/bin/env python3
"""psacx_construct_u64 on host pointers (SURVEY 8(d) Metric 1: what psac's own timer spans, src/psac.cpp:95-121), phase by phase:
tools/host_path_time.py [log2 n = 32] [calls = 3].  Random DNA made on the device; the first call pays for the page faults of the
result arrays and is not shown."""
import os as _os; _os.environ.setdefault("PSACX_ENV_KNOBS", "1")      # PSACX_* variables select the forms of single stages (psac_amd/_lib.py: ENV_KNOBS)
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import psac_amd

logn = int(sys.argv[1]) if len(sys.argv) > 1 else 32
calls = int(sys.argv[2]) if len(sys.argv) > 2 else 3
n = 1 << logn
bits = 64 if n > (1 << 31) else 32
ctx = psac_amd.Context(0)
d_text = ctx.alloc(n)
ctx.check(ctx._lib.psacx_synth_text_dev(ctx.handle, d_text, n, 0, 0, 1, 0))
text = np.empty(n, np.uint8)
ctx.d2h(text, d_text); ctx.free(d_text)
sa = psac_amd.SuffixArray(index_bits=bits, lcp=True, ctx=ctx)
sa.construct(text)
names = ("text up", "construction", "rest of SA + LCP (early)", "ISA (+ SA, LCP) down", "Lc down", "call")
for it in range(calls):
    t0 = time.perf_counter()
    sa.local_SA, sa.local_B, sa.local_LCP = sa.construct_into(text, sa.local_SA, sa.local_B, sa.local_LCP)
    dt = time.perf_counter() - t0
    ms = list(ctx.stats().ms_host)
    print("2^%d random DNA, uint%d, host pointers: %.1f ms = %.2f GChars/s;  " % (logn, bits, dt * 1e3, n / dt / 1e9)
          + ", ".join("%s %.1f" % (names[i], ms[i]) for i in range(6))
          + ("; early: began %.1f, SA + LCP through %.1f" % (ms[6], ms[8]) if ms[6] > 0 else ""), flush=True)
