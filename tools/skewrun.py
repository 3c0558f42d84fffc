#!/usr/bin/env python3
"""Non-uniform inputs at scale: tools/skewrun.py <log2 n> <bits>.  Builds (a) a low-entropy text (geometric
symbol frequencies over 20 symbols), (b) a text of repeated reads with mutations (long shared prefixes), (c) random DNA with
40 % interspersed repeats (1000 families of 300 bp, 5 % divergence),
constructs SA+ISA+LCP on the GPU, verifies on the device, prints timings."""
import os as _os; _os.environ.setdefault("PSACX_ENV_KNOBS", "1")      # PSACX_* variables select the forms of single stages (psac_amd/_lib.py: ENV_KNOBS)
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import psac_amd

logn = int(sys.argv[1]); bits = int(sys.argv[2])
n = 1 << logn
w = bits // 8
rng = np.random.RandomState(7)


def geometric_text(n):
    p = 0.5 ** np.arange(1, 21); p /= p.sum()
    return (97 + rng.choice(20, size=n, p=p)).astype(np.uint8)


def mutated_reads(n):
    base = np.frombuffer(b"ACGT", np.uint8)[rng.randint(0, 4, size=1 << 16)]
    reps = n // base.size
    t = np.tile(base, reps + 1)[:n].copy()
    mut = rng.randint(0, n, size=n // 200)               # one mutation every 200 characters
    t[mut] = np.frombuffer(b"ACGT", np.uint8)[rng.randint(0, 4, size=mut.size)]
    return t


def interspersed_repeats(n, frac=0.4):
    """random DNA of which `frac` is covered by copies of 1000 repeat families of 300 bp, 5 % mutations per copy (a genome's
    interspersed repeats: a quarter of the suffixes tie on the sorted prefix, in groups of up to 10^5)"""
    r = np.random.RandomState(11)
    acgt = np.frombuffer(b"ACGT", np.uint8)
    t = acgt[r.randint(0, 4, size=n)].copy()
    fam = acgt[r.randint(0, 4, size=(1000, 300))]
    copies = int(n * frac / 300)
    pos = r.randint(0, max(1, n - 300), size=copies)
    which = r.randint(0, 1000, size=copies)
    for a in range(0, copies, 1 << 16):
        blk = fam[which[a:a + (1 << 16)]].copy()
        mut = r.rand(*blk.shape) < 0.05
        blk[mut] = acgt[r.randint(0, 4, size=int(mut.sum()))]
        idx = pos[a:a + (1 << 16), None] + np.arange(300)[None, :]
        t[idx.ravel()] = blk.ravel()
    return t


ctx = psac_amd.Context(0)
for name, gen in (("geometric20", geometric_text), ("mutated_repeats", mutated_reads), ("interspersed_repeats_40", interspersed_repeats)):
    text = gen(n)
    d_text = ctx.alloc(n); ctx.h2d(d_text, text)
    d_sa, d_isa, d_lcp = ctx.alloc(n * w), ctx.alloc(n * w), ctx.alloc(n * w)
    sa = psac_amd.SuffixArray(index_bits=bits, lcp=True, ctx=ctx)
    for it in range(2):
        t0 = time.time()
        s = sa.construct_device(d_text, n, d_sa, d_isa, d_lcp, profile=True)
        dt = time.time() - t0
    err = psac_amd.check_device(ctx, d_text, n, d_sa, d_isa, d_lcp, bits)
    print(json.dumps({"input": name, "n": n, "bits": bits, "seconds": round(dt, 4), "MChars_per_s": round(n / dt / 1e6, 1),
                      "rounds": [(r.h, r.unfinished_buckets, r.unfinished_elements, r.sort_passes) for r in s.rounds[:s.n_rounds]],
                      "check_errors": err,
                      "phases_ms": {"keys": round(s.ms_kmer, 2), "scatter": round(s.ms_sort_scatter + s.ms_sort_scatter3 + s.ms_sort_scatter2, 2),
                                    "tilehist": round(s.ms_sort_tilehist, 2), "rebucket": round(s.ms_rebucket, 2), "isa": round(s.ms_isa_scatter, 2),
                                    "gather": round(s.ms_gather, 2), "compact": round(s.ms_compact, 2)}}), flush=True)
    for p in (d_text, d_sa, d_isa, d_lcp):
        ctx.free(p)
