#!/usr/bin/env python3
"""A/B of an environment switch inside one process (same memory placement): tools/ab_env.py ENVNAME [log2 n] [bits]."""
import os as _os; _os.environ.setdefault("PSACX_ENV_KNOBS", "1")      # PSACX_* variables select the forms of single stages (psac_amd/_lib.py: ENV_KNOBS)
import ctypes as C
import os
import sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import psac_amd
env = sys.argv[1]; logn = int(sys.argv[2]) if len(sys.argv) > 2 else 32; bits = int(sys.argv[3]) if len(sys.argv) > 3 else 64
n = 1 << logn; w = bits // 8
ctx = psac_amd.Context(0)
lib = ctx._lib if hasattr(ctx, "_lib") else psac_amd._lib.load()
d_text = ctx.alloc(n)
assert lib.psacx_synth_text_dev(ctx.handle, C.c_void_p(d_text), n, 0, 0, 1, 1024) == 0
d_sa, d_isa, d_lcp = ctx.alloc(n * w), ctx.alloc(n * w), ctx.alloc(n * w)
sa = psac_amd.SuffixArray(index_bits=bits, lcp=True, ctx=ctx)
sa.construct_device(d_text, n, d_sa, d_isa, d_lcp)
for rep in range(3):
    for on in (False, True):
        if on: os.environ[env] = os.environ.get("AB_VALUE", "1")
        else: os.environ.pop(env, None)
        s = sa.construct_device(d_text, n, d_sa, d_isa, d_lcp, profile=True)
        print("%s=%d: keys %.1f, tile hist %.1f, scatter %.1f, ties %.2f, rebucket %.2f, isa %.1f, total %.1f"
              % (env, on, s.ms_kmer, s.ms_sort_tilehist, s.ms_sort_scatter + s.ms_sort_scatter3 + s.ms_sort_scatter2, s.ms_gather, s.ms_rebucket, s.ms_isa_scatter, s.ms_total))
