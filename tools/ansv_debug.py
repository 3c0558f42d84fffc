#!/usr/bin/env python3
"""Small ANSV cases against the oracle, printing the first mismatches (debug aid)."""
import os as _os; _os.environ.setdefault("PSACX_ENV_KNOBS", "1")      # PSACX_* variables select the forms of single stages (psac_amd/_lib.py: ENV_KNOBS)
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import oracle_lib as O
import psac_amd
ctx = psac_amd.Context(0)
rng = np.random.default_rng(17)
NO = 2**64 - 1
cases = [rng.integers(0, 100, n).astype(np.uint32) for n in (1, 2, 13, 137, 1000, 26666)]
cases.append(rng.integers(0, 3, 70000).astype(np.uint64))
cases.append(np.zeros(5000, np.uint32))
cases.append(np.arange(5000, dtype=np.uint32))
cases.append(np.arange(5000, dtype=np.uint64)[::-1].copy())
bad = 0
for ci, v in enumerate(cases):
    for lt in (0, 1, 2):
        for rt in (0, 1, 2):
            left, right = psac_amd.ansv(v, lt, rt, nonsv=NO, ctx=ctx)
            for name, got, ref in (("L", left, O.ansv(v, True, lt, NO)), ("R", right, O.ansv(v, False, rt, NO))):
                w = np.nonzero(got != ref)[0]
                if w.size:
                    bad += 1
                    print("case %d n=%d lt=%d rt=%d side %s: %d wrong; first:" % (ci, v.size, lt, rt, name, w.size),
                          [(int(i), int(v[i]), int(got[i]) if got[i] != NO else -1, int(ref[i]) if ref[i] != NO else -1) for i in w[:6]])
print("mismatching (case, pair, side):", bad)
