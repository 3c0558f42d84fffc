#!/usr/bin/env python3
"""Randomised parity run on the GPU: tools/fuzz.py <seconds> [seed].  Random lengths (1 .. 2^23, biased to the
thresholds), alphabets, generators, index widths, k, fast_resolval, LCP / Lc / string-set modes; every result is
compared with the CPU oracle (SA, ISA, LCP, Lc, round log).  Stops at the first mismatch."""
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

budget = float(sys.argv[1]) if len(sys.argv) > 1 else 60.0
seed = int(sys.argv[2]) if len(sys.argv) > 2 else 1
rng = np.random.RandomState(seed)
ctx = psac_amd.Context(0)
t_end = time.time() + budget
runs = 0
refused = 0


def pick_n():
    r = rng.rand()
    if r < 0.25:
        return int(rng.randint(1, 300))
    if r < 0.5:
        return int(rng.randint(300, 100000))
    if r < 0.75:
        return int((1 << 21) + rng.randint(-3, 4) * rng.randint(0, 3))
    return int(rng.randint(1 << 20, 1 << 23))


def make_text(n):
    kind = rng.randint(0, 6)
    sigma = int(rng.choice([1, 2, 3, 4, 5, 20, 64, 127, 200, 256]))
    lo = int(rng.randint(0, 257 - sigma))
    if kind == 0:
        return (lo + rng.randint(0, sigma, size=n)).astype(np.uint8), "uniform s=%d" % sigma
    if kind == 1:
        p = 0.5 ** np.arange(1, sigma + 1); p /= p.sum()
        return (lo + rng.choice(sigma, size=n, p=p)).astype(np.uint8), "geometric s=%d" % sigma
    if kind == 2:
        per = int(rng.randint(1, 5000))
        unit = (lo + rng.randint(0, sigma, size=per)).astype(np.uint8)
        return np.tile(unit, n // per + 1)[:n].copy(), "periodic %d s=%d" % (per, sigma)
    if kind == 3:
        base = (lo + rng.randint(0, sigma, size=max(1, n // 7))).astype(np.uint8)
        t = np.tile(base, 8)[:n].copy()
        if n > 10:
            mut = rng.randint(0, t.size, size=max(1, n // 500))
            t[mut] = (lo + rng.randint(0, sigma, size=mut.size)).astype(np.uint8)
        return t, "mutated repeats s=%d" % sigma
    if kind == 4:
        t = (lo + rng.randint(0, sigma, size=n)).astype(np.uint8)
        t[n - min(n, int(rng.randint(1, 60))):] = lo          # tail of the smallest symbol
        return t, "min-tail s=%d" % sigma
    return np.full(n, lo, np.uint8), "constant"


while time.time() < t_end:
    n = pick_n()
    text, what = make_text(n)
    bits = int(rng.choice([32, 64]))
    mode = rng.randint(0, 10)
    desc = "n=%d %s uint%d" % (n, what, bits)
    if mode == 0 and n >= 2:                                       # string set
        m = int(rng.randint(1, min(n, 2000) + 1))
        cuts = np.unique(np.concatenate([[0, n], rng.randint(1, n, size=m - 1)])) if m > 1 else np.array([0, n])
        strings = [bytes(text[a:b]) for a, b in zip(cuts[:-1], cuts[1:])]
        got = psac_amd.SuffixArray(index_bits=bits, lcp=True, ctx=ctx)
        try:
            got.construct_ss(strings)
        except psac_amd.PsacxError as e:
            if os.environ.get("PSACX_DIET_CAP") and "larger than the reduced-memory layout" in str(e):
                refused += 1
                continue
            raise
        ref = O.construct_ss(strings, bits=bits)
        ok = np.array_equal(got.local_SA, ref["SA"]) and np.array_equal(got.local_B, ref["ISA"]) and np.array_equal(got.local_LCP, ref["LCP"])
        desc += " string set of %d" % len(strings)
    elif n == 1:
        got = psac_amd.SuffixArray(index_bits=bits, lcp=True, ctx=ctx); got.construct(text)
        ok = got.local_SA.tolist() == [0] and got.local_LCP.tolist() == [0]
    else:
        fast = bool(rng.rand() < 0.8)
        l = max(1, int(np.unique(text).size).bit_length())
        k = 0 if rng.rand() < 0.7 else int(rng.randint(1, max(2, bits // l + 1)))
        lc = bool(rng.rand() < 0.2)
        lcp = lc or bool(rng.rand() < 0.8)
        desc += " fast=%s k=%d lcp=%s lc=%s" % (fast, k, lcp, lc)
        got = psac_amd.SuffixArray(index_bits=bits, lcp=lcp, lc=lc, ctx=ctx)
        try:
            got.construct(text, fast_resolval=fast, k=k)
        except psac_amd.PsacxError as e:
            # (with PSACX_DIET_CAP set: a bucket of unresolved suffixes beyond the forced capacity is a refusal, not a wrong answer)
            if os.environ.get("PSACX_DIET_CAP") and "larger than the reduced-memory layout" in str(e):
                refused += 1
                continue
            raise
        if k and k >= n:
            k = 0 if n < 2 else k
        ref = O.construct_lc(text, bits=bits, fast=fast, k=k) if lc else O.construct(text, bits=bits, fast=fast, k=k, lcp=lcp)
        ok = np.array_equal(got.local_SA, ref["SA"]) and np.array_equal(got.local_B, ref["ISA"])
        if lcp:
            ok = ok and np.array_equal(got.local_LCP, ref["LCP"])
        if lc:
            ok = ok and np.array_equal(got.local_Lc, ref["Lc"])
    runs += 1
    if not ok:
        print("MISMATCH:", desc, flush=True)
        np.save("/tmp/fuzz_fail.npy", text)
        sys.exit(1)
print("fuzz: %d runs in %.0f s, all equal to the oracle (seed %d)%s" % (runs, budget, seed, ", %d refused for the forced capacity" % refused if refused else ""))
