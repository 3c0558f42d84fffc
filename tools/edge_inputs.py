#!/usr/bin/env python3
"""Alphabet and size edge cases around the 2^21 threshold (two-stage first round, three-kernel passes) against the
CPU oracle, on the GPU: tools/edge_inputs.py.  SA, ISA, LCP, k and the per-round log must all agree."""
import os as _os; _os.environ.setdefault("PSACX_ENV_KNOBS", "1")      # PSACX_* variables select the forms of single stages (psac_amd/_lib.py: ENV_KNOBS)
import sys, os
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import psac_amd, oracle_lib as O, inputs
ctx = psac_amd.Context(0)
rng = np.random.RandomState(3)
cases = []
n0 = 1 << 21
for n in (n0 - 1, n0, n0 + 1, 3 * n0 + 17):
    cases.append(("bytes256 n=%d" % n, rng.randint(0, 256, size=n).astype(np.uint8)))
    cases.append(("sigma2 n=%d" % n, (rng.randint(0, 2, size=n) + 65).astype(np.uint8)))
cases.append(("protein20 2^22", (rng.randint(0, 20, size=1 << 22) + 65).astype(np.uint8)))
cases.append(("sigma3 skew 2^22", (65 + rng.choice(3, size=1 << 22, p=[0.9, 0.09, 0.01])).astype(np.uint8)))
cases.append(("zeros+ones bytes 2^21", rng.randint(0, 2, size=n0 + 9).astype(np.uint8)))
bad = 0
for name, text in cases:
    for bits in (32, 64):
        sa = psac_amd.SuffixArray(index_bits=bits, lcp=True, ctx=ctx); sa.construct(text)
        ref = O.construct(text, bits=bits)
        ok = np.array_equal(sa.local_SA, ref["SA"]) and np.array_equal(sa.local_B, ref["ISA"]) and np.array_equal(sa.local_LCP, ref["LCP"])
        ok = ok and sa.k == ref["k"] and [(h, b, e) for h, b, e, *_ in sa.rounds] == [(h, b, e) for h, b, e, _ in ref["trace"]]
        print(name, bits, "OK" if ok else "MISMATCH", flush=True)
        bad += not ok
print("bad", bad)
