#!/usr/bin/env python3
"""Randomised parity run of the ANSV kernel on the GPU: tools/fuzz_ansv.py <seconds> [seed].  Arrays long enough that a wave works
through many tiles (2^22 .. 2^26 elements: the answers a wave carries from tile to tile, psac_amd/csrc/ansv_wave.hpp step 6, are what
small arrays never exercise), of shapes chosen to stress them: few / many distinct values at tile edges, falling and rising runs longer
than a tile, plateaus across tiles, rare deep minima, LCP arrays.  Every (left_type, right_type) result is compared with the oracle's
ansv (ansv.hpp:48-65 restated).  Stops at the first mismatch."""
import os as _os; _os.environ.setdefault("PSACX_ENV_KNOBS", "1")      # PSACX_* variables select the forms of single stages (psac_amd/_lib.py: ENV_KNOBS)
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import inputs
import oracle_lib as O
import psac_amd


def make(rng, n, kind):
    """kind -> (array of n values as uint64, description)"""
    if kind == 0:
        r = int(rng.choice([2, 3, 5, 17, 100, 10**6]))
        return rng.randint(0, r, size=n).astype(np.uint64), "uniform in [0, %d)" % r
    if kind == 1:          # saw teeth: falling or rising runs of length L (a falling run longer than a tile: every element asks beyond it)
        L = int(rng.choice([50, 1000, 1500, 5000, 100000]))
        i = np.arange(n, dtype=np.uint64) % np.uint64(L)
        return (np.uint64(L) - i if rng.rand() < 0.5 else i), "saw teeth of %d" % L
    if kind == 2:          # plateaus: runs of one value, mean length M, levels from a small or a large range
        M = int(rng.choice([3, 300, 3000, 40000]))
        cuts = np.flatnonzero(rng.rand(n) < 1.0 / M)
        lev = rng.randint(0, int(rng.choice([3, 40, 10**5])), size=cuts.size + 1).astype(np.uint64)
        return lev[np.searchsorted(cuts, np.arange(n), side="right")], "plateaus of about %d" % M
    if kind == 3:          # LCP-like: a narrow band of values, with rare deep minima that answer elements many tiles away
        v = (10 + rng.geometric(0.35, size=n)).astype(np.uint64)
        deep = np.flatnonzero(rng.rand(n) < float(rng.choice([1e-3, 1e-4, 1e-5])))
        v[deep] = rng.randint(0, 10, size=deep.size).astype(np.uint64)
        return v, "narrow band with %d deep minima" % deep.size
    if kind == 4:          # many distinct values at tile edges: a level per 1024-tile plus noise
        lev = (np.arange(n, dtype=np.uint64) // np.uint64(int(rng.choice([256, 1024, 4096])))) * np.uint64(2654435761) % np.uint64(int(rng.choice([23, 40, 97])))
        return lev * np.uint64(3) + rng.randint(0, 4, size=n).astype(np.uint64), "a level per tile"
    v = rng.randint(0, 6, size=n).astype(np.uint64)                              # long stretches without small values, then a zero
    z = np.flatnonzero(rng.rand(n) < 1e-6); v += np.uint64(1); v[z] = 0
    return v, "small values, %d zeros" % z.size


def main():
    budget = float(sys.argv[1]) if len(sys.argv) > 1 else 60.0
    seed = int(sys.argv[2]) if len(sys.argv) > 2 else 1
    rng = np.random.RandomState(seed)
    ctx = psac_amd.Context(0)
    t_end = time.time() + budget
    runs = 0
    NO = (1 << 64) - 1
    lcp_cache = {}
    while time.time() < t_end:
        logn = int(rng.choice([22, 23, 24, 25, 26], p=[0.2, 0.2, 0.3, 0.2, 0.1]))
        n = (1 << logn) + int(rng.choice([0, 0, 1, -1, 777, -12345]))
        bits = int(rng.choice([32, 64]))
        kind = int(rng.randint(0, 7))
        if kind == 6:      # a real LCP array (random DNA, or a text of repeats with mutations)
            key = (min(logn, 24), int(rng.randint(0, 2)))
            if key not in lcp_cache:
                m = 1 << key[0]
                text = inputs.dna(m, 3) if key[1] == 0 else np.tile(inputs.dna(m // 64, 4), 64)
                if key[1] == 1:
                    mut = rng.randint(0, m, size=m // 300); text = text.copy(); text[mut] = inputs.dna(mut.size, 9)
                sa = psac_amd.SuffixArray(index_bits=32, lcp=True, ctx=ctx); sa.construct(text)
                lcp_cache[key] = np.asarray(sa.local_LCP).astype(np.uint64)
            v, what = lcp_cache[key], "LCP array (%s)" % ("random DNA" if key[1] == 0 else "mutated repeats")
            n = v.size
        else:
            v, what = make(rng, n, kind)
        v = v.astype(np.uint32 if bits == 32 else np.uint64)
        w = bits // 8
        d_in, d_l, d_r = ctx.alloc(n * w), ctx.alloc(n * 8), ctx.alloc(n * 8)
        ctx.h2d(d_in, v)
        want = {}
        pairs = [(2, 0)] + [(int(rng.randint(0, 3)), int(rng.randint(0, 3)))]
        for lt, rt in pairs:
            psac_amd.ansv_device(ctx, d_in, n, d_l, d_r, bits, lt, rt, NO)
            L = np.empty(n, np.uint64); R = np.empty(n, np.uint64)
            ctx.d2h(L, d_l); ctx.d2h(R, d_r)
            for side, t, got in ((True, lt, L), (False, rt, R)):
                if (side, t) not in want:
                    want[(side, t)] = O.ansv(v, side, t, NO)
                if not np.array_equal(got, want[(side, t)]):
                    bad = np.flatnonzero(got != want[(side, t)])
                    print("MISMATCH: n %d uint%d %s, types (%d, %d), %s side: %d wrong, first at %d (tile %d): got %d want %d  [seed %d run %d]"
                          % (n, bits, what, lt, rt, "left" if side else "right", bad.size, bad[0], bad[0] // 1024, got[bad[0]], want[(side, t)][bad[0]], seed, runs))
                    sys.exit(1)
            runs += 1
        for p in (d_in, d_l, d_r):
            ctx.free(p)
    print("fuzz_ansv: %d calls in %.0f s, all equal to the oracle (seed %d)" % (runs, budget, seed))


if __name__ == "__main__":
    main()
