#!/usr/bin/env python3
"""Randomised parity run of the refinement rounds of LONG buckets (rank requests through partition levels, heavy / light split:
psac_amd/csrc/heavy_keys.hpp): tools/fuzz_long.py <seconds> [seed].  Periodic texts of 2^23 .. 2^25 characters (period 1 .. 5000, some with a
few substitutions, some with a random tail), uint64, SA + ISA + LCP, both layouts and rounds in slabs; every result is compared with the
construction that fetches every rank on its own and sorts all records (PSACX_GATHER=fetch, PSACX_NO_HEAVY=1: the path the oracle suite
covers), the round log too where no round ran in slabs.  Stops at the first mismatch."""
import os as _os; _os.environ.setdefault("PSACX_ENV_KNOBS", "1")      # PSACX_* variables select the forms of single stages (psac_amd/_lib.py: ENV_KNOBS)
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import psac_amd

budget = float(sys.argv[1]) if len(sys.argv) > 1 else 60.0
seed = int(sys.argv[2]) if len(sys.argv) > 2 else 1
rng = np.random.RandomState(seed)
ctx = psac_amd.Context(0)
KNOBS = ("PSACX_GATHER", "PSACX_NO_HEAVY", "PSACX_FORCE_DIET", "PSACX_DIET_CAP", "PSACX_NO_WHOLE", "PSACX_ISA_UPDATE", "PSACX_NO_LAZY_RANKS")


def construct(text, env):
    for k in KNOBS:
        os.environ.pop(k, None)
    os.environ.update(env)
    n = int(text.size)
    d_text = ctx.alloc(n); ctx.h2d(d_text, text)
    d = [ctx.alloc(n * 8) for _ in range(3)]
    sa = psac_amd.SuffixArray(index_bits=64, lcp=True, ctx=ctx)
    st = sa.construct_device(d_text, n, d[0], d[1], d[2])
    out = [np.empty(n, np.uint64) for _ in range(3)]
    for a, p in zip(out, d):
        ctx.d2h(a, p)
    for p in [d_text] + d:
        ctx.free(p)
    return out, st, [r[:3] for r in sa.rounds]


t_end = time.time() + budget
runs = heavy = refused = 0
while time.time() < t_end:
    n = int(rng.randint((1 << 23) + 1, 1 << 25))
    per = int(rng.choice([1, 2, 3, 7, 64, 100, 1024, 1500, 4097, int(rng.randint(1, 5000))]))
    sigma = int(rng.choice([2, 4, 20]))
    unit = (65 + rng.randint(0, sigma, size=per)).astype(np.uint8)
    text = np.tile(unit, n // per + 1)[:n].copy()
    what = "n=%d period %d sigma %d" % (n, per, sigma)
    r = rng.rand()
    if r < 0.4:
        m = int(rng.randint(1, 200))
        text[rng.randint(0, n, m)] = 65 + sigma
        what += " %d substitutions" % m
    elif r < 0.6:
        tl = int(rng.randint(1, n // 4))
        text[n - tl:] = (65 + rng.randint(0, sigma, size=tl)).astype(np.uint8)
        what += " random tail %d" % tl
    env = {}
    if rng.rand() < 0.5:
        env["PSACX_FORCE_DIET"] = "1"
        if rng.rand() < 0.5:
            env["PSACX_DIET_CAP"] = str(int(rng.randint(5 << 20, n)))
    if rng.rand() < 0.2:
        env["PSACX_NO_HEAVY"] = "1"
    if rng.rand() < 0.2:
        env["PSACX_ISA_UPDATE"] = "stores"
    if rng.rand() < 0.15:
        env["PSACX_NO_LAZY_RANKS"] = "1"
    try:
        (SA, ISA, LCP), st, log = construct(text, env)
    except psac_amd.PsacxError as e:
        # (a bucket of unresolved suffixes beyond the forced capacity is a refusal, not a wrong answer)
        if "PSACX_DIET_CAP" in env and "larger than the reduced-memory layout" in str(e):
            refused += 1
            continue
        raise
    (rSA, rISA, rLCP), _, rlog = construct(text, {"PSACX_GATHER": "fetch", "PSACX_NO_HEAVY": "1"})
    ok = np.array_equal(SA, rSA) and np.array_equal(ISA, rISA) and np.array_equal(LCP, rLCP)
    if "PSACX_DIET_CAP" not in env:
        ok = ok and log == rlog
    runs += 1
    heavy += int(st.heavy_rounds > 0)
    if not ok:
        print("MISMATCH:", what, env, flush=True)
        np.save("/tmp/fuzz_long_fail.npy", text)
        sys.exit(1)
print("fuzz_long: %d constructions compared (%d refused: a bucket beyond the forced capacity), %d of them with split rounds, seed %d: all equal" % (runs, refused, heavy, seed))
