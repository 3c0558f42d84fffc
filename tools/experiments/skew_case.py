#!/usr/bin/env python3
"""tools/experiments/skew_case.py <log2 n> <case index 0..2>: one input of tools/skewrun.py, uint64, two constructions (for rocprofv3)."""
import os as _os; _os.environ.setdefault("PSACX_ENV_KNOBS", "1")      # PSACX_* variables select the forms of single stages (psac_amd/_lib.py: ENV_KNOBS)
import os, sys, runpy
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import psac_amd
src = open(os.path.join(ROOT, "tools", "skewrun.py")).read().split("ctx = psac_amd.Context(0)")[0]
sys.argv = [sys.argv[0], sys.argv[1], "64", sys.argv[2]]
ns = {"__file__": os.path.join(ROOT, "tools", "skewrun.py")}
exec(compile(src, "skewrun_head", "exec"), ns)
n = ns["n"]; gen = [ns["geometric_text"], ns["mutated_reads"], ns["interspersed_repeats"]][int(sys.argv[3])]
t = gen(n)
ctx = psac_amd.Context(0)
d_text = ctx.alloc(n); ctx.h2d(d_text, t)
d = [ctx.alloc(n * 8) for _ in range(3)]
sa = psac_amd.SuffixArray(index_bits=64, lcp=True, ctx=ctx)
for it in range(2):
    s = sa.construct_device(d_text, n, d[0], d[1], d[2], profile=True)
print("total %.1f ms" % s.ms_total)
