#!/usr/bin/env python3
"""tools/experiments/multi_genome_like.py <P> <log2 n>: the interspersed-repeats text of tools/skewrun.py on P virtual ranks (host path),
against the one-GPU engine."""
import os as _os; _os.environ.setdefault("PSACX_ENV_KNOBS", "1")      # PSACX_* variables select the forms of single stages (psac_amd/_lib.py: ENV_KNOBS)
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import psac_amd
P = int(sys.argv[1]); logn = sys.argv[2]
src = open(os.path.join(ROOT, "tools", "skewrun.py")).read().split("ctx = psac_amd.Context(0)")[0]
sys.argv = [sys.argv[0], logn, "64"]
ns = {"__file__": os.path.join(ROOT, "tools", "skewrun.py")}
exec(compile(src, "skewrun_head", "exec"), ns)
t = ns["interspersed_repeats"](ns["n"])
ctx = psac_amd.Context(0)
one = psac_amd.SuffixArray(index_bits=64, lcp=True, ctx=ctx); one.construct(t)
mg = psac_amd.MultiContext([0] * P)
for it in range(2):
    t0 = time.time(); SA, ISA, LCP, rounds = mg.construct(t, index_bits=64); dt = time.time() - t0
print("P=%d 2^%s: %.1f ms (host path), equal to one GPU: %s %s %s; forms %s; rounds %d" % (P, logn, dt * 1e3, np.array_equal(SA, one.local_SA),
      np.array_equal(ISA, one.local_B), np.array_equal(LCP, one.local_LCP), mg.last_form(), len(rounds)))
mg.close()
