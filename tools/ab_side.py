"""Wall time of construct_device on the side workloads of bench.py (for A/B runs of a knob): python tools/ab_side.py KIND LOG2N PERIOD [REPS] [check]
KIND: 0 random DNA, 2 tandem repeat, 3 repeated reads with mutations (psacx_synth_text_dev).  check: the result of the last construction goes
through the recurrence checker of the multi-GPU engine on one rank (LCP by ranks, not by characters: repetitive texts)."""
import os as _os; _os.environ.setdefault("PSACX_ENV_KNOBS", "1")      # PSACX_* variables select the forms of single stages (psac_amd/_lib.py: ENV_KNOBS)
import sys, time, ctypes as C, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import psac_amd
kind, lg, period = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3])
reps = int(sys.argv[4]) if len(sys.argv) > 4 else 3
n = 1 << lg
ctx = psac_amd.Context(0)
d_text = ctx.alloc(n)
ctx.check(ctx._lib.psacx_synth_text_dev(ctx.handle, C.c_void_p(d_text), n, 0, kind, 7, period))
d_sa, d_isa, d_lcp = ctx.alloc(n * 8), ctx.alloc(n * 8), ctx.alloc(n * 8)
sa = psac_amd.SuffixArray(index_bits=64, lcp=True, ctx=ctx)
sa.construct_device(d_text, n, d_sa, d_isa, d_lcp)
ts = []
for _ in range(reps):
    t0 = time.perf_counter()
    st = sa.construct_device(d_text, n, d_sa, d_isa, d_lcp)
    ts.append((time.perf_counter() - t0) * 1e3)
err = None
if len(sys.argv) > 5 and sys.argv[5] == "check":
    ctx.check(ctx._lib.psacx_trim(ctx.handle))
    mgc = psac_amd.MultiContext([ctx.device])
    try:
        err = list(mgc.check_device([d_text], [n], [d_sa], [d_isa], [d_lcp], 64))
    finally:
        mgc.close()
print("kind", kind, "n 2^%d" % lg, "ms", " ".join("%.1f" % t for t in ts), "rounds", st.n_rounds, "check", err)
print("rounds (h, records, unfinished buckets, unfinished elements, sort passes):", [(r.h, r.active, r.unfinished_buckets, r.unfinished_elements, r.sort_passes) for r in st.rounds[:st.n_rounds]])
