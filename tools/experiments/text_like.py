#!/usr/bin/env python3
"""tools/experiments/text_like.py <log2 n>: words of a Zipf vocabulary (50000 words of 2..12 lower-case letters) joined by spaces;
SA + ISA + LCP, uint64, phases + device check."""
import os as _os; _os.environ.setdefault("PSACX_ENV_KNOBS", "1")      # PSACX_* variables select the forms of single stages (psac_amd/_lib.py: ENV_KNOBS)
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import psac_amd
n = 1 << int(sys.argv[1])
rng = np.random.RandomState(5)
V = 50000
lens = rng.randint(2, 13, size=V)
letters = rng.choice(26, size=int(lens.sum()), p=np.array([8.2,1.5,2.8,4.3,12.7,2.2,2.0,6.1,7.0,0.15,0.77,4.0,2.4,6.7,7.5,1.9,0.095,6.0,6.3,9.1,2.8,0.98,2.4,0.15,2.0,0.074])/100.0*100/ (np.array([8.2,1.5,2.8,4.3,12.7,2.2,2.0,6.1,7.0,0.15,0.77,4.0,2.4,6.7,7.5,1.9,0.095,6.0,6.3,9.1,2.8,0.98,2.4,0.15,2.0,0.074]).sum()))
starts = np.concatenate([[0], np.cumsum(lens)])
p = 1.0 / np.arange(1, V + 1); p /= p.sum()
avg = float((lens * p).sum()) + 1
cnt = int(n / avg * 1.1)
ws = rng.choice(V, size=cnt, p=p)
wl = lens[ws] + 1
ends = np.cumsum(wl)
keep = int(np.searchsorted(ends, n))
ws = ws[:keep + 1]; wl = wl[:keep + 1]
total = int(wl.sum())
t = np.full(total, 32, np.uint8)
off = np.concatenate([[0], np.cumsum(wl)[:-1]])
for L in range(2, 13):
    sel = np.nonzero(lens[ws] == L)[0]
    if sel.size == 0: continue
    src = starts[ws[sel]][:, None] + np.arange(L)[None, :]
    dst = off[sel][:, None] + np.arange(L)[None, :]
    t[dst.ravel()] = (97 + letters[src.ravel()]).astype(np.uint8)
t = t[:n].copy()
ctx = psac_amd.Context(0)
d_text = ctx.alloc(n); ctx.h2d(d_text, t)
d = [ctx.alloc(n * 8) for _ in range(3)]
sa = psac_amd.SuffixArray(index_bits=64, lcp=True, ctx=ctx)
for it in range(2):
    t0 = time.time()
    s = sa.construct_device(d_text, n, d[0], d[1], d[2], profile=True)
    dt = time.time() - t0
err = psac_amd.check_device(ctx, d_text, n, d[0], d[1], d[2], 64)
print("text-like 2^%d: %.1f ms wall, total %.1f ms: keys %.1f scatter %.1f tilehist %.1f rebucket %.1f isa %.1f gather(ties) %.1f compact %.1f rmq %.1f; k %d; rounds %s; check %s" % (
    int(sys.argv[1]), dt * 1e3, s.ms_total, s.ms_kmer, s.ms_sort_scatter + s.ms_sort_scatter2 + s.ms_sort_scatter3, s.ms_sort_tilehist, s.ms_rebucket,
    s.ms_isa_scatter, s.ms_gather, s.ms_compact, s.ms_rmq_build, sa.k, [(r[0], r[1], r[2]) for r in sa.rounds][:12], err))
