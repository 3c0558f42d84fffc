#!/usr/bin/env python3
"""Randomised parity run of the multi-GPU engine (psacx_multi_*, ranks sharing device 0): tools/fuzz_multi.py <seconds> [seed].
Random rank counts (1 .. 8), lengths (a few characters per rank .. 2^21), alphabets, generators (uniform, geometric, periodic,
repeats with mutations, one symbol), index widths, both layouts and small slabs (PSACX_MULTI_SLAB picked at random so that
refinement rounds run in several steps); SA, ISA, LCP and the per-round log are compared with the CPU oracle.  Stops at the
first mismatch."""
import os as _os; _os.environ.setdefault("PSACX_ENV_KNOBS", "1")      # PSACX_* variables select the forms of single stages (psac_amd/_lib.py: ENV_KNOBS)
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import oracle_lib as O
import psac_amd

budget = float(sys.argv[1]) if len(sys.argv) > 1 else 60.0
seed = int(sys.argv[2]) if len(sys.argv) > 2 else 1
rng = np.random.RandomState(seed)
t_end = time.time() + budget
runs = 0


def make_text(n):
    kind = rng.randint(0, 5)
    sigma = int(rng.choice([1, 2, 4, 5, 20, 64, 200]))
    lo = int(rng.randint(0, 257 - sigma))
    if kind == 0:
        return (lo + rng.randint(0, sigma, size=n)).astype(np.uint8), "uniform s=%d" % sigma
    if kind == 1:
        p = 0.5 ** np.arange(1, sigma + 1); p /= p.sum()
        return (lo + rng.choice(sigma, size=n, p=p)).astype(np.uint8), "geometric s=%d" % sigma
    if kind == 2:
        per = int(rng.randint(1, 3000))
        unit = (lo + rng.randint(0, sigma, size=per)).astype(np.uint8)
        return np.tile(unit, n // per + 1)[:n].copy(), "periodic %d s=%d" % (per, sigma)
    if kind == 3:
        base = (lo + rng.randint(0, sigma, size=max(1, n // 9))).astype(np.uint8)
        t = np.tile(base, 10)[:n].copy()
        mut = rng.randint(0, t.size, size=max(1, n // 300))
        t[mut] = (lo + rng.randint(0, sigma, size=mut.size)).astype(np.uint8)
        return t, "mutated repeats s=%d" % sigma
    return np.full(n, lo, np.uint8), "constant"


while time.time() < t_end:
    P = int(rng.randint(1, 9))
    r = rng.rand()
    n = int(rng.randint(P * 3, 4000)) if r < 0.3 else int(rng.randint(4000, 300000)) if r < 0.8 else int(rng.randint(300000, 1 << 21))
    text, what = make_text(n)
    bits = int(rng.choice([32, 64]))
    layout = int(rng.choice([1, 2]))
    slab = int(rng.choice([0, 0, 64, 1000, 20000]))
    desc = "P=%d n=%d %s uint%d layout=%d slab=%d" % (P, n, what, bits, layout, slab)
    mg = psac_amd.MultiContext([0] * P)
    try:
        mg.configure(layout=layout, slab=slab)
        SA, ISA, LCP, rounds = mg.construct(text, index_bits=bits)
    finally:
        mg.close()
    ref = O.construct(text, bits=bits)
    bad = [w for w, a, b in (("SA", SA, ref["SA"]), ("ISA", ISA, ref["ISA"]), ("LCP", LCP, ref["LCP"])) if not np.array_equal(a, b)]
    # (the log is compared where it must agree with a run on one rank: no slabs -- their counters may run ahead -- and blocks longer than a
    #  word's characters, below which psac clamps k to the shortest block, kmer.hpp:26-40)
    if not bad and slab == 0 and layout == 1 and n // P > 70 and rounds != [(h, b, e) for h, b, e, _ in ref["trace"]]:
        bad = ["round log %s vs %s" % (rounds, [(h, b, e) for h, b, e, _ in ref["trace"]])]
    ok = not bad
    if bad: desc += " -- differs in " + ", ".join(bad)
    runs += 1
    if not ok:
        print("MISMATCH:", desc, flush=True)
        np.save("/tmp/fuzz_multi_fail.npy", text)
        sys.exit(1)
print("fuzz_multi: %d runs in %.0f s, all equal to the oracle (seed %d)" % (runs, budget, seed))
