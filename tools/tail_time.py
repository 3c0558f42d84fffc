#!/usr/bin/env python3
"""Launch-bound refinement rounds: tools/tail_time.py [log2 n] [period] [bits].  A tandem repeat keeps every suffix
unresolved for log2(n / period) rounds of n records each; with n = 2^20 a round is ~20 us of work.  Prints ms per
construction and per refinement round."""
import os as _os; _os.environ.setdefault("PSACX_ENV_KNOBS", "1")      # PSACX_* variables select the forms of single stages (psac_amd/_lib.py: ENV_KNOBS)
import os
import sys
import time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import inputs
import psac_amd

logn = int(sys.argv[1]) if len(sys.argv) > 1 else 20
period = int(sys.argv[2]) if len(sys.argv) > 2 else 256
bits = int(sys.argv[3]) if len(sys.argv) > 3 else 32
n = 1 << logn
w = bits // 8
ctx = psac_amd.Context(0)
for name, text in (("tandem(2^%d, %d)" % (logn, period), inputs.tandem(n, period, inputs.dna(period, 3))),
                   ("one symbol 2^%d" % logn, np.full(n, 65, np.uint8))):
    d_text = ctx.alloc(n); ctx.h2d(d_text, text)
    d_sa, d_isa, d_lcp = ctx.alloc(n * w), ctx.alloc(n * w), ctx.alloc(n * w)
    sa = psac_amd.SuffixArray(index_bits=bits, lcp=True, ctx=ctx)
    best = 1e9
    for it in range(5):
        t0 = time.perf_counter()
        s = sa.construct_device(d_text, n, d_sa, d_isa, d_lcp)
        best = min(best, time.perf_counter() - t0)
    # (the device checker compares characters: linear in sum(LCP), minutes for long repeats above 2^22 characters;
    # there the multi-GPU engine's checker, which verifies LCP through its recurrence, runs with one rank)
    if logn <= 22:
        err = psac_amd.check_device(ctx, d_text, n, d_sa, d_isa, d_lcp, bits)
    else:
        mg = psac_amd.MultiContext([0])
        err = mg.check_device([d_text], [n], [d_sa], [d_isa], [d_lcp], bits)
        mg.close()
    print("%s uint%d: %.3f ms, %d rounds -> %.3f ms per round; check %s" % (name, bits, best * 1e3, s.n_rounds, best * 1e3 / s.n_rounds, err))
    s = sa.construct_device(d_text, n, d_sa, d_isa, d_lcp, profile=True)
    print("   phases of one construction (ms, HIP events): keys %.1f, sort hist %.1f, tile hist %.1f, scatter %.1f, rebucket %.1f, isa %.1f, gather %.1f, compact %.1f, rmq %.1f, total %.1f"
          % (s.ms_kmer, s.ms_sort_hist, s.ms_sort_tilehist, s.ms_sort_scatter + s.ms_sort_scatter3 + s.ms_sort_scatter2, s.ms_rebucket, s.ms_isa_scatter, s.ms_gather,
             s.ms_compact, s.ms_rmq_build, s.ms_total))
    for p in (d_text, d_sa, d_isa, d_lcp):
        ctx.free(p)
