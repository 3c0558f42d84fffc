#!/bin/bash
# rocprofv3 kernel trace of one construction of 2^30 characters of repeated reads with mutations (psacx_synth_text_dev kind 3), uint64
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/$1
mkdir -p $OUT
cat > /tmp/mut.py <<PY
import sys, time, ctypes as C
sys.path.insert(0, "$R")
import psac_amd
n = 1 << 30
ctx = psac_amd.Context(0)
d_text = ctx.alloc(n)
ctx.check(ctx._lib.psacx_synth_text_dev(ctx.handle, C.c_void_p(d_text), n, 0, 3, 7, 1 << 16))
d_sa, d_isa, d_lcp = ctx.alloc(n * 8), ctx.alloc(n * 8), ctx.alloc(n * 8)
sa = psac_amd.SuffixArray(index_bits=64, lcp=True, ctx=ctx)
sa.construct_device(d_text, n, d_sa, d_isa, d_lcp)
t0 = time.perf_counter()
st = sa.construct_device(d_text, n, d_sa, d_isa, d_lcp, profile=True)
print("ms", (time.perf_counter() - t0) * 1e3, "rounds", st.n_rounds, [(r.h, r.active, r.unfinished_buckets, r.sort_passes) for r in st.rounds[:st.n_rounds]])
PY
rocprofv3 --kernel-trace --stats -d /tmp/$1 -o m -- python /tmp/mut.py > $OUT/run.log 2>&1
python3 $R/tools/rocpd_summary.py /tmp/$1/m_results.db > $OUT/trace_summary.txt 2>&1
grep -a "^ms" $OUT/run.log; head -32 $OUT/trace_summary.txt | cut -c1-170
