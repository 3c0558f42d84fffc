#!/usr/bin/env python3
"""Large single-GPU runs: tools/bigrun.py <n> <bits> <dna|ascii128> [steps].  Builds the text in chunks,
constructs SA+ISA+LCP with everything resident in HBM, verifies on the device, prints timings."""
import os as _os; _os.environ.setdefault("PSACX_ENV_KNOBS", "1")      # PSACX_* variables select the forms of single stages (psac_amd/_lib.py: ENV_KNOBS)
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import inputs
import psac_amd

n = int(eval(sys.argv[1])); bits = int(sys.argv[2]); kind = sys.argv[3]; steps = int(sys.argv[4]) if len(sys.argv) > 4 else 1
w = bits // 8
ctx = psac_amd.Context(0)
d_text = ctx.alloc(n)
CH = 1 << 28
t0 = time.time()
for o in range(0, n, CH):
    m = min(CH, n - o)
    z = inputs.splitmix64_stream(m, 1 + o)          # chunk-seeded stream
    blk = (np.frombuffer(b"ACGT", np.uint8)[(z & np.uint64(3)).astype(np.int64)] if kind == "dna" else (z & np.uint64(127)).astype(np.uint8))
    ctx.h2d(d_text + o, blk)
print("text ready in %.1f s" % (time.time() - t0), flush=True)
d_sa, d_isa, d_lcp = ctx.alloc(n * w), ctx.alloc(n * w), ctx.alloc(n * w)
sa = psac_amd.SuffixArray(index_bits=bits, lcp=True, ctx=ctx)
for it in range(steps):
    t0 = time.time()
    s = sa.construct_device(d_text, n, d_sa, d_isa, d_lcp, profile=True)
    dt = time.time() - t0
    q = max(range(3), key=lambda i: s.scatter_bytes[i])       # dominant form of the scatter kernel
    ms = (s.ms_sort_scatter, s.ms_sort_scatter3, s.ms_sort_scatter2)[q]
    print(json.dumps({"n": n, "bits": bits, "kind": kind, "seconds": round(dt, 3), "MChars_per_s": round(n / dt / 1e6, 1),
                      "k": s.k, "rounds": [(r.h, r.unfinished_buckets, r.unfinished_elements, r.sort_passes) for r in s.rounds[:s.n_rounds]],
                      "scatter_form": ("look-back", "three-word", "two-word")[q], "scatter_launches": s.scatter_launches[q],
                      "scatter_ms_per_pass": round(ms / max(1, s.scatter_launches[q]), 3),
                      "scatter_GBps": round(s.scatter_bytes[q] / (ms * 1e-3) / 1e9, 1) if ms else None,
                      "workspace_GiB": round(s.workspace_bytes / 2**30, 1),
                      "phases_ms": {"keys": round(s.ms_kmer, 1), "scatter": round(s.ms_sort_scatter + s.ms_sort_scatter3 + s.ms_sort_scatter2, 1), "tilehist": round(s.ms_sort_tilehist, 1),
                                    "rebucket": round(s.ms_rebucket, 1), "isa": round(s.ms_isa_scatter, 1), "compact": round(s.ms_compact, 1),
                                    "rmq": round(s.ms_rmq_build, 1), "gather": round(s.ms_gather, 1)}}), flush=True)
t0 = time.time()
err = psac_amd.check_device(ctx, d_text, n, d_sa, d_isa, d_lcp, bits)
print("device check errors:", err, "in %.1f s" % (time.time() - t0), flush=True)
