#!/usr/bin/env python3
"""Times the C++ multi-GPU engine with P ranks sharing device 0 (virtual ranks, copy transport):
tools/multi_time.py <P> <log2 n total> <bits> [kind].  Reports ms per construction and the exchange volume."""
import os as _os; _os.environ.setdefault("PSACX_ENV_KNOBS", "1")      # PSACX_* variables select the forms of single stages (psac_amd/_lib.py: ENV_KNOBS)
import ctypes as C
import sys
import time
import os
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import psac_amd

P = int(sys.argv[1]); n = 1 << int(sys.argv[2]); bits = int(sys.argv[3]); kind = sys.argv[4] if len(sys.argv) > 4 else "dna"
w = bits // 8
mg = psac_amd.MultiContext([0] * P)
lib = mg._lib
sizes = [n // P + (1 if r < n % P else 0) for r in range(P)]
offs = [sum(sizes[:r]) for r in range(P)]
d = dict(text=[], sa=[], isa=[], lcp=[])
for r in range(P):
    ctx = mg.rank_ctx(r)
    def alloc(nb):
        p = C.c_void_p(); assert lib.psacx_dev_alloc(ctx, C.byref(p), nb) == 0; return p.value
    d["text"].append(alloc(sizes[r]))
    assert lib.psacx_synth_text_dev(ctx, C.c_void_p(d["text"][r]), sizes[r], offs[r], {"dna": 0, "ascii128": 1, "tandem": 2}[kind], 1, 1024) == 0
    for key in ("sa", "isa", "lcp"):
        d[key].append(alloc((sizes[r] + sizes[0] // 8 + 256) * w))      # (with the slack that lets the reduced-memory layout use them as record arrays)
mg.configure(output_slack=sizes[0] // 8 + 256)
for it in range(3):
    t0 = time.perf_counter()
    st, sent, ex, ga = mg.construct_device(d["text"], sizes, d["sa"], d["isa"], d["lcp"], bits)
    dt = time.perf_counter() - t0
print("P=%d n=2^%s uint%d %s: %.2f ms per construction = %.1f MChars/s; rounds %d; %d exchanges, %d scalar gathers, %.2f GB moved between ranks"
      % (P, sys.argv[2], bits, kind, dt * 1e3, n / dt / 1e6, st.n_rounds, ex, ga, sent / 1e9))
print("phases (host wall ms):", "; ".join("%s %.1f" % (k, v) for k, v in mg.phases()))
print("exchange ms on the second streams:", mg.wire()["exchange_ms"], mg.last_form())
err = mg.check_device(d["text"], sizes, d["sa"], d["isa"], d["lcp"], bits)
print("distributed check errors:", err)
mg.close()
