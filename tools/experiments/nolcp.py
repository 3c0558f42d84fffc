import sys, time, ctypes as C, os
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import psac_amd
n = 1 << 30
ctx = psac_amd.Context(0)
d_text = ctx.alloc(n)
ctx.check(ctx._lib.psacx_synth_text_dev(ctx.handle, C.c_void_p(d_text), n, 0, 3, 7, 1 << 16))
d_sa, d_isa, d_lcp = ctx.alloc(n * 8), ctx.alloc(n * 8), ctx.alloc(n * 8)
for lcp in (True, False):
    sa = psac_amd.SuffixArray(index_bits=64, lcp=lcp, ctx=ctx)
    sa.construct_device(d_text, n, d_sa, d_isa, d_lcp if lcp else None)
    t0 = time.perf_counter()
    st = sa.construct_device(d_text, n, d_sa, d_isa, d_lcp if lcp else None, profile=True)
    print("lcp", lcp, "ms %.1f" % ((time.perf_counter() - t0) * 1e3), "rebucket %.1f gather %.1f scatter %.1f isa %.1f compact %.1f rmq %.1f kmer %.1f hist %.1f" % (st.ms_rebucket, st.ms_gather, st.ms_sort_scatter + st.ms_sort_scatter2 + st.ms_sort_scatter3, st.ms_isa_scatter, st.ms_compact, st.ms_rmq_build, st.ms_kmer, st.ms_sort_tilehist))
