#!/usr/bin/env python3
"""Footprint and time of the distributed construction per rank, the ranks sharing device 0 (virtual ranks):
  tools/dist_bigrun.py <P> <log2 characters per rank> <bits> <kind: dna|ascii128|tandem> <layout: normal|reduced> [slab]
(BIGRUN_EXTRA=e adds e characters to every block: 2 ranks of 2^31 + e characters are a text of more than 2^32 characters
on one GPU, i.e. the 64-bit payload forms of the exchanges; BIGRUN_ITERS=k times the k-th construction)
Prints the device memory every rank's engine allocated at its peak in words per character (beside the three result
arrays and the text the caller owns), the total against the 288 GB of one MI355X for a block of 2^32 characters with
64-bit words (BASELINE.json configs[4]: 32 GiB over 8 GPUs), ms per construction and the distributed checker's verdict."""
import os as _os; _os.environ.setdefault("PSACX_ENV_KNOBS", "1")      # PSACX_* variables select the forms of single stages (psac_amd/_lib.py: ENV_KNOBS)
import ctypes as C
import os
import sys
import time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import psac_amd

P = int(sys.argv[1]); m = (1 << int(sys.argv[2])) + int(os.environ.get("BIGRUN_EXTRA", "0")); bits = int(sys.argv[3]); kind = sys.argv[4]; layout = sys.argv[5]
slab = int(sys.argv[6]) if len(sys.argv) > 6 else 0
w = bits // 8
n = m * P
mg = psac_amd.MultiContext([0] * P)
lib = mg._lib
slack = m // 8 + 256
sizes = [m] * P
d = dict(text=[], sa=[], isa=[], lcp=[])
for r in range(P):
    ctx = mg.rank_ctx(r)
    def alloc(nb):
        p = C.c_void_p(); assert lib.psacx_dev_alloc(ctx, C.byref(p), nb) == 0; return p.value
    d["text"].append(alloc(m))
    assert lib.psacx_synth_text_dev(ctx, C.c_void_p(d["text"][r]), m, r * m, {"dna": 0, "ascii128": 1, "tandem": 2}[kind], 3 if kind == "tandem" else 1, 1024) == 0
    for key in ("sa", "isa", "lcp"):
        d[key].append(alloc((m + slack) * w))
mg.configure(layout={"normal": 1, "reduced": 2, "auto": 0}[layout], slab=slab, output_slack=slack)
for it in range(int(os.environ.get("BIGRUN_ITERS", "1"))):      # the last one is reported
    t0 = time.perf_counter()
    st, sent, ex, ga = mg.construct_device(d["text"], sizes, d["sa"], d["isa"], d["lcp"], bits)
    dt = time.perf_counter() - t0
phases, form, wire = mg.phases(), mg.last_form(), mg.wire()
peak, reduced, slab_rounds = mg.memory()
words = max(peak) / float(m * w)
own = 3.0 * (m + slack) / m + 1.0 / w
print("P=%d, 2^%s characters per rank, uint%d, %s, layout %s%s: %.1f ms, %d rounds (%d in slabs), %.2f GB between ranks"
      % (P, sys.argv[2], bits, kind, "reduced" if reduced else "normal", (", slab %d" % slab) if slab else "", dt * 1e3, st.n_rounds, slab_rounds, sent / 1e9))
print("  engine allocations at their peak: %.2f words per character (max over ranks); with the results (3 x %.3f) and the text: %.2f words"
      % (words, (m + slack) / float(m), words + own))
print("  a block of 2^32 characters with 64-bit words would hold %.1f GB of the 288 GB of an MI355X" % ((words + own) * 8 * 2.0 ** 32 / 1e9))
err = mg.check_device(d["text"], sizes, d["sa"], d["isa"], d["lcp"], bits)
print("  phases (host wall ms):", ", ".join("%s %.1f" % (k.strip(), v) for k, v in phases))
print("  forms:", form, " wire:", wire)
print("  distributed check errors:", err)
mg.close()
sys.exit(0 if err == [0, 0, 0, 0] else 1)
