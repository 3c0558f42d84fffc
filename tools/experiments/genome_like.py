#!/usr/bin/env python3
"""tools/experiments/genome_like.py <log2 n> [fraction]: random DNA of which `fraction` (default 0.4) is covered by copies of 1000
repeat families of 300 bp with 5 % mutations per copy (interspersed repeats); SA + ISA + LCP, uint64, phases + device check."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import psac_amd
n = 1 << int(sys.argv[1]); frac = float(sys.argv[2]) if len(sys.argv) > 2 else 0.4
rng = np.random.RandomState(11)
acgt = np.frombuffer(b"ACGT", np.uint8)
t = acgt[rng.randint(0, 4, size=n)].copy()
fam = acgt[rng.randint(0, 4, size=(1000, 300))]
copies = int(n * frac / 300)
pos = rng.randint(0, n - 300, size=copies)
which = rng.randint(0, 1000, size=copies)
CH = 1 << 16
for a in range(0, copies, CH):
    p = pos[a:a + CH]; w = which[a:a + CH]
    blk = fam[w].copy()
    mut = rng.rand(*blk.shape) < 0.05
    blk[mut] = acgt[rng.randint(0, 4, size=int(mut.sum()))]
    idx = p[:, None] + np.arange(300)[None, :]
    t[idx.ravel()] = blk.ravel()
ctx = psac_amd.Context(0)
d_text = ctx.alloc(n); ctx.h2d(d_text, t)
d = [ctx.alloc(n * 8) for _ in range(3)]
sa = psac_amd.SuffixArray(index_bits=64, lcp=True, ctx=ctx)
for it in range(2):
    t0 = time.time()
    s = sa.construct_device(d_text, n, d[0], d[1], d[2], profile=True)
    dt = time.time() - t0
err = psac_amd.check_device(ctx, d_text, n, d[0], d[1], d[2], 64)
print("genome-like 2^%d, %.0f %% repeats: %.1f ms wall, total %.1f ms: keys %.1f scatter %.1f tilehist %.1f rebucket %.1f isa %.1f gather(ties) %.1f compact %.1f rmq %.1f; rounds %s; check %s" % (
    int(sys.argv[1]), frac * 100, dt * 1e3, s.ms_total, s.ms_kmer, s.ms_sort_scatter + s.ms_sort_scatter2 + s.ms_sort_scatter3, s.ms_sort_tilehist, s.ms_rebucket,
    s.ms_isa_scatter, s.ms_gather, s.ms_compact, s.ms_rmq_build, [(r[0], r[1], r[2]) for r in sa.rounds][:12], err))
