#!/usr/bin/env python3
"""bench.py -- SA+LCP construction throughput of the HIP engine on MI355X.

One "step" = one full suffix-array + inverse-SA + LCP construction of the
synthetic text, text already resident in HBM, results left in HBM.

N = 1 workload: the north-star's headline shape, a 4 GiB random DNA string
(sigma = 4, splitmix64 seed 1, n = 2^32), uint64 indices, SA + ISA + LCP on one
MI355X (the largest single-GPU configuration; BASELINE.json configs[2], 4 GiB
random ASCII, is `--alphabet ascii128`; configs[1], 256 MiB DNA with uint32
indices, is `--n 268435456 --index 32`).  The text is generated in HBM
(psacx_synth_text_dev), the result of the last timed step is verified in HBM
(psacx_check_dev_*), and the PCIe-inclusive construct() time on host pointers
(SURVEY 8(d) Metric 1) is reported beside `value` as `construct_host`.

Prints ONE JSON line (rank 0): metric MChars/s, plus
  roofline     -- the dominant kernel (the radix scatter pass of the rank-pair
                  sort): algorithmic bytes (6w per record per pass)
                  / HIP-event time of those launches inside the timed region
  cpu_baseline -- the CPU oracle (a port of psac's algorithm, 1 thread) timed on
                  this box's host on a bounded sample of the same kind of text.
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

HBM_PEAK_GBS = 8000.0          # MI355X_MICROARCH.md: HBM3E 8 TB/s spec
# HBM bytes per launch of the dominant kernel from the PMC counters of the committed profile
# (profiles/): FETCH_SIZE doubled (the gfx950 correction for coalesced streams) + WRITE_SIZE.
# Valid for the default workload only (2^28 uint32 records per launch).
# [1] three-word form, profiles/r01_pmc_*.txt: (2 x 7454089 + 17090295) KiB over 5 launches
# [2] two-word form, profiles/r01d_pmc_*.txt: (2 x 5903929.2 + 12713084.0) KiB over 6 launches
# [2] at 2^32 uint64 records (the default workload; 32-bit payload between the passes), profiles/r02w_pmc_*.txt: (2 x 120112432.6 + 279687304.6) KiB over 5 launches = 106478012375
# ... round 3 (three workgroups per CU under a register cap: the spilled registers are written through), profiles/r03s_pmc_*.txt:
#     per launch (2 x 25644637.9 + 63871449.6) KiB for passes 1-4, (2 x 25699635.6 + 81472582.0) KiB for pass 5
# ... end of round 3, profiles/r04m_pmc_*.txt: (2 x 25644777.8 + 64191682.1) KiB for passes 1-4, (2 x 27795142.8 + 84983547.1) KiB
#     for pass 5; mean over the five passes = 123391751168 bytes
# ... after the scratch of the narrow passes went from 20 to 4 bytes per thread, profiles/r04s_pmc_*.txt: (2 x 24602539.1 + 59913695.6) KiB
#     for passes 1-4, (2 x 26743427.4 + 82173525.6) KiB for pass 5; mean = 117173345280 bytes (1.14 x the algorithmic bytes: runs of
#     64-128 bytes that start anywhere write whole 32-byte sectors)
# ... the prefix sort in one-word records (form 3), profiles/r04v_pmc_*.txt: top-digit pass (2 x 17320582.1 + 36377016.7) KiB, bucket passes
#     (2 x 17323107.6 + 34451517.1) KiB x 3, widening pass (2 x 17328677.1 + 76887497.3) KiB; mean over the five = 79842555904 bytes
#     (1.056 x the algorithmic 17.6 bytes per record and pass)
# ... with the fused front end (the pass on the top digit no longer counted here), profiles/r04y_pmc_*.txt: bucket passes
#     (2 x 17324186.8 + 34509922.4) KiB x 3, widening pass (2 x 17330780.3 + 78369040.8) KiB; mean over the four = 82049404928 bytes
#     (1.061 x the algorithmic 18 bytes per record and pass)
# ... round 4, profiles/r4b_bench_{fetch,write}_4gib_u64.txt: bucket passes (2 x 17324028 + 34444799) KiB x 3, widening pass
#     (2 x 17328092 + 76522428) KiB; mean over the four = 81525037312 bytes (1.054 x the algorithmic 18 bytes per record and pass)
# ... round 4, later (no widening pass any more), profiles/r4d_bench_{fetch,write}_4gib_u64.txt: four bucket passes of
#     (2 x 17323970 + 34463887) KiB = 70770510848 bytes each (1.030 x the algorithmic 16 bytes per record and pass)
# Keyed by (scatter form, records per launch, index bits).
# ... round 4, last code, profiles/r4f_bench_{fetch,write}_4gib_u64.txt: (2 x 17324001 + 34476460) KiB = 70783449088 bytes (1.030 x)
# ... round 5 (match-any ranking in four instructions per bit), profiles/r5f_bench_{fetch,write}_4gib_u64.txt: (2 x 17324315 + 35328914) KiB = 71657005056 bytes (1.043 x)
# ... round 5, last code (the digit byte of the next pass written beside the record by three of the four passes, padded fronts of the pass on the top digit),
#     profiles/r5j_bench_{fetch,write}_4gib_u64.txt: (2 x 17325130 + 40009069) KiB = 76451152896 bytes (1.063 x the algorithmic 16.75 bytes per record and pass)
# ... round 5, final code (every bucket pass writes a byte: three the next digit, the last the byte the tie stage reads),
#     profiles/r5n_bench_{fetch,write}_4gib_u64.txt: (2 x 17325395 + 41356537) KiB = 77831502848 bytes (1.066 x the algorithmic 17 bytes per record and pass)
# ... round 6 (same kernels; the run of the final code), profiles/r6_bench_{fetch,write}_4gib_u64.txt: (2 x 17325391 + 41437855) KiB = 77914764288 bytes (1.067 x)
# Every entry names the committed profile it was read from (roofline.traffic_profile).
TRAFFIC = {(1, 1 << 28, 32): (6553287372, "profiles/r01_pmc_fetch_size.txt + r01_pmc_write_size.txt"),
           (2, 1 << 28, 32): (4184907503, "profiles/r01d_pmc_fetch_size.txt + r01d_pmc_write_size.txt"),
           (2, 1 << 32, 64): (117173345280, "profiles/r04s_pmc_fetch_size_4gib_u64.txt + r04s_pmc_write_size_4gib_u64.txt"),
           (3, 1 << 32, 64): (77914764288, "profiles/r6_bench_fetch_4gib_u64.txt + r6_bench_write_4gib_u64.txt")}


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--n", "--size", dest="n", type=int, default=None, help="characters per GPU (default: 2^32 -- BASELINE.json configs[3] is 32 GiB over 8 GPUs; with more than one GPU 2^28 if a GPU has less than ~250 GB free)")
    ap.add_argument("--index", type=int, default=None, choices=(32, 64), help="index width (default: 64 above 2^31 characters)")
    ap.add_argument("--alphabet", default="dna", choices=("dna", "ascii128", "tandem"))
    ap.add_argument("--seed", type=int, default=1)
    ap.add_argument("--cpu-sample", type=int, default=1 << 28, help="characters for the CPU baseline leg (0 = skip); the default is the whole workload, ~8 s on the GPU box's 256 host threads")
    ap.add_argument("--no-lcp", action="store_true")
    ap.add_argument("--host-path", default="auto", choices=("auto", "full", "off"),
                    help="also time psacx_construct_* with host pointers (H2D of the text, D2H of SA/ISA/LCP: SURVEY 8(d) "
                         "Metric 1); reported as the extra field construct_host, never as `value`.  auto: the full workload "
                         "when the host has the memory for its results, else 2^28 characters")
    ap.add_argument("--no-check", action="store_true", help="skip the device checker on the last result")
    ap.add_argument("--side", default="auto", choices=("auto", "off"),
                    help="after the timed region of the default workload also time BASELINE.json configs[2] (4 GiB ASCII, uint64), "
                         "configs[1] (256 MiB DNA, uint32) and the three-word (B1,B2,idx) form of the scatter pass; reported as "
                         "the extra field other_workloads, never as `value`")
    a = ap.parse_args()
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if a.n is None and world == 1 and a.gpus == 1 and not os.environ.get("PSACX_BENCH_FORCE_DIST"):
        a.n = 1 << 32
    return a


def make_text(kind, n, seed):
    import inputs
    if kind == "dna":
        return inputs.dna(n, seed)
    if kind == "ascii128":
        return inputs.ascii128(n, seed)
    return inputs.tandem(n, 1024, inputs.dna(1024, seed))


def cpu_baseline(kind, sample, seed, bits):
    """psac's algorithm restated on the CPU (oracle/psac_ref.cpp), built with OpenMP loops and the parallel-mode
    sort and run on all host cores (the reference is an MPI code that uses every core it is given)."""
    import oracle_lib as O
    text = make_text(kind, sample, seed)
    cores = os.cpu_count() or 1
    os.environ.setdefault("OMP_NUM_THREADS", str(cores))
    t0 = time.perf_counter()
    O.construct_all_cores(text, bits=bits)
    dt = time.perf_counter() - t0
    out = {"value": round(sample / dt / 1e6, 3), "unit": "MChars/s", "cores": cores, "kind": "port",
           "sample": "%d chars of the same generator (%s, seed %d), SA+LCP, uint%d, %.1f s on %d threads"
                     % (sample, kind, seed, bits, dt, cores)}
    # the port's single-core throughput beside the survey's figure for psac itself on one core of its VM
    # (16 MiB random DNA, uint64, k = 21: 6631 ms = 2.53 MChars/s; uint32: 4.06) -- the same 2^24 characters
    try:
        ps = min(sample, 1 << 24)
        t0 = time.perf_counter()
        O.construct(make_text("dna", ps, seed), bits=bits)
        t1 = time.perf_counter() - t0
        ref = 2.53 if bits == 64 else 4.06
        out["port_1_thread"] = {"value": round(ps / t1 / 1e6, 3), "unit": "MChars/s", "cores": 1, "kind": "port",
                                "sample": "%d chars of random DNA (seed %d), SA+ISA+LCP, uint%d, %.1f s on 1 thread" % (ps, seed, bits, t1),
                                "survey_psac_1_core_MChars_per_s": ref, "ratio_to_survey_psac": round(ps / t1 / 1e6 / ref, 2),
                                "note": "BASELINE.md section 2 timed psac's own headers on one core of the survey VM (Xeon 2.1 GHz) on the same "
                                        "generator family and size; this is the oracle's restatement on one core of this box"}
    except Exception as e:          # a side measurement never breaks the bench line
        out["port_1_thread"] = {"error": str(e)[:200]}
    # SURVEY 8(d) CPU baseline leg (1): libdivsufsort (the reference's own CPU comparison, src/psac_vs_dss.cpp:
    # 87-119; oracle/_ref build of /root/reference/ext/libdivsufsort) + Kasai LCP, one thread, bounded sample
    if O.have_divsufsort():
        ds = min(sample, 1 << 26)
        t0 = time.perf_counter()
        SA = O.divsufsort(text[:ds], bits)
        t1 = time.perf_counter()
        O.kasai(text[:ds], SA, O.inverse(SA))
        t2 = time.perf_counter()
        out["divsufsort"] = {"value": round(ds / (t2 - t0) / 1e6, 3), "unit": "MChars/s", "cores": 1, "kind": "reference",
                             "sample": "%d chars, divsufsort%s %.1f s + inverse/Kasai %.1f s, 1 thread"
                                       % (ds, "64" if bits == 64 else "", t1 - t0, t2 - t1)}
    return out


def report(a, world, n, bits, dt, scat_ms, scat_bytes, scat_launches, phases, k, l, rounds, parallelism, onew_passes=0, scat_records=None):
    w = bits // 8
    ms_per_step = dt / a.steps * 1e3
    value = world * n * a.steps / dt / 1e6
    dom = max(range(len(scat_bytes)), key=lambda q: scat_bytes[q])
    kname = ("radix_scatter_kernel (one 8-bit digit pass of the (B1,B2,idx) rank-pair sort, look-back form)",
             "radix_scatter3_kernel (one 8-bit digit pass of the (B1,B2,idx) rank-pair sort)",
             "radix_scatter3_kernel<two-word> (one 8-bit digit pass of the first round's (B1,idx) prefix sort)")[dom]
    achieved = scat_bytes[dom] / (scat_ms[dom] * 1e-3) / 1e9 if scat_ms[dom] > 0 else 0.0
    tkey = dom
    if dom == 2 and onew_passes > 0:         # (psacx_stats.onew_passes: bucket passes over one-word records ran)
        # the prefix sort in one-word records (engine.hpp: prefix_sort_1w): the passes inside the buckets
        # (radix_scatter1w_kernel<..., 8>: 8 + 8 bytes per record) and the last one (radix_scatter1w_kernel<..., 9>: one word in,
        # word 1 + suffix out); the figures below are their mean
        kname = ("radix_scatter1w_kernel (one 8-bit digit pass of the first round's prefix sort in one-word records, 8 + 8 bytes per record and one "
                 "byte more: the digit the next pass sorts on, which its tile histograms read instead of the records -- from the last pass the lowest "
                 "byte of the prefix, in which the tie stage looks for its groups instead of reading the records; the rebucket kernel reads the "
                 "records where the last pass leaves them; the pass on the top digit computes its keys from the text and is timed with them)")
        tkey = 3
    out = {
        "metric": "MChars/s SA+LCP build; rank-pair radix-sort HBM GB/s vs peak",
        "value": round(value, 2), "unit": "MChars/s", "n_gpus": world, "steps": a.steps,
        "warmup": a.warmup, "ms_per_step": round(ms_per_step, 3), "higher_is_better": True,
        "scaling": "weak", "vs_baseline": None, "dtype": "u%d" % bits, "data": "synthetic",
        "config": {"workload": "%d MiB %s per GPU (%s), %d MiB in total, uint%d indices, "
                               "SA+%s on %d x MI355X" % (n >> 20, {"dna": "random DNA (sigma 4)", "ascii128": "random ASCII (sigma 128)",
                                                                   "tandem": "period-1024 tandem repeat of random DNA"}[a.alphabet],
                                                         ("splitmix64 seed %d" % a.seed) if world == 1 else
                                                         ("rank r holds block r of one splitmix64 stream, seed %d" % a.seed),
                                                         (world * n) >> 20, bits,
                                                         "ISA" if a.no_lcp else "ISA+LCP", world),
                   "n_per_gpu": n, "k": k, "bits_per_char": l, "rounds": rounds, "parallelism": parallelism,
                   "value_definition": "characters / construction time with the text resident in HBM and SA, ISA, LCP left in HBM; "
                                       "SURVEY 8(d) Metric 1 (construct() on host pointers, PCIe copies included) is the field "
                                       "construct_host, never `value`"},
        "roofline": {"bound": "hbm", "kernel": kname,
                     "achieved": round(achieved, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                     "frac": round(achieved / HBM_PEAK_GBS, 4),
                     "avg_launch_ms": round(scat_ms[dom] / max(scat_launches[dom], 1), 4),
                     "launches_per_step": scat_launches[dom] // max(a.steps, 1),
                     "algorithmic_bytes_per_launch": scat_bytes[dom] // max(scat_launches[dom], 1),
                     "records_per_launch": (scat_records[dom] // max(scat_launches[dom], 1)) if scat_records else n,
                     "bytes_per_record_per_pass": round(scat_bytes[dom] / float(max(scat_records[dom], 1)), 2) if scat_records
                                                  else round(scat_bytes[dom] / max(scat_launches[dom], 1) / float(n), 2),
                     "traffic": TRAFFIC.get((tkey, n, bits), (None, None))[0] if world == 1 else None,
                     "traffic_profile": TRAFFIC.get((tkey, n, bits), (None, None))[1] if world == 1 else None,
                     "traffic_source": "PMC counters (FETCH_SIZE doubled per the gfx950 correction + WRITE_SIZE, separate passes) of the committed "
                                       "profile named in traffic_profile, per launch of this kernel on this workload; not measured in this run"},
    }
    if phases:
        out["phase_ms_last_step"] = phases
    return out


def main_distributed(a, rank, world, local_rank):
    """N > 1: one process per GPU.  The text (one stream DNA(world * n, seed), ...) is block-partitioned over the
    ranks like psac partitions it over MPI ranks; the construction is the C++ multi-GPU engine behind
    psacx_multi_* (psac_amd/csrc/multi.hpp): sample-sort shuffle, SA->ISA scatter, B2 fetch and range-min queries
    as grouped ncclSend / ncclRecv over xGMI on a second stream.  torch.distributed only carries the communicator
    id to the ranks and brackets the timed region."""
    import ctypes as C
    # RCCL prints a version banner on the C stdout of every process: while the libraries run, file descriptor 1 is the
    # process's stderr, and it is the JSON line alone that goes to the real stdout at the end
    sys.stdout.flush()
    real_stdout = os.dup(1)
    os.dup2(2, 1)
    import torch
    import torch.distributed as dist
    import psac_amd
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    os.environ.setdefault("MASTER_PORT", "29533")
    os.environ.setdefault("RANK", str(rank)); os.environ.setdefault("WORLD_SIZE", str(world))
    torch.cuda.set_device(local_rank)
    dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))
    if a.n is None:
        # The largest block of 2^32 (BASELINE.json configs[4]: 32 GiB over 8 GPUs) or 2^31 (configs[3]: 16 GiB over 8 GPUs) characters per
        # GPU that fits every GPU.  The reduced-memory layout of the multi-GPU engine peaks at 2.63 words per character for the engine
        # (profiles/r4d_*: the one-word first round keeps its records in the rank's result arrays, the re-balance runs in place) + 3.375
        # for the result arrays with their slack + the text = 6.13 words of 8 bytes: 196 GiB at 2^32 characters.  Asked for: 7 words + 8 GiB
        # (RCCL's buffers, the block cache's odd sizes).  A GPU that cannot hold 2^31 is not the machine this benchmark is defined on:
        # the run stops with a reason instead of quoting a smaller workload as if it were the configured one (--n overrides).
        # Random text: 2^32 per GPU only while the whole text has at most 2^34 characters (N <= 4) -- the one-word records of the first
        # round keep 64 - bits_for(n - 1) bits of prefix beside the suffix, and beyond 2^34 characters a quarter of the suffixes of a random
        # text tie on them and take the slower tie stage; eight GPUs take 2^31 each: BASELINE.json configs[3], 16 GiB of DNA over 8 GPUs.
        # The tandem repeat (configs[4]: 32 GiB over 8 GPUs) takes 2^32 per GPU at every N: all its suffixes tie on any prefix anyway, the
        # reduced-memory layout orders them slab by slab (multi.hpp: first_sort_ties) and peaks below 3 engine words per character
        # (profiles/r5a_multi_tandem_*: one rank x 2^32 2.48, 8 x 2^28 2.92; 8.25 and "does not fit" in round 4).
        free_b = torch.cuda.mem_get_info(local_rank)[0]
        fit = 0
        for lg in (32, 31):
            if world * (1 << lg) > (1 << 34) and a.alphabet != "tandem":
                continue
            if free_b >= int(7.0 * 8 * (1 << lg)) + (8 << 30):
                fit = lg
                break
        t_fit = torch.tensor([fit], dtype=torch.int32, device="cuda")
        dist.all_reduce(t_fit, op=dist.ReduceOp.MIN)
        if int(t_fit.item()) == 0:
            if rank == 0:
                sys.stderr.write("bench.py --gpus %d: a rank has %.1f GiB of free device memory, fewer than the %.1f GiB a block of 2^31 characters "
                                 "needs (7 words of 8 bytes per character + 8 GiB); pass --n to run a smaller block\n"
                                 % (world, free_b / 2.0 ** 30, (7.0 * 8 * (1 << 31) + (8 << 30)) / 2.0 ** 30))
            dist.destroy_process_group()
            os.dup2(real_stdout, 1)
            sys.exit(3)
        a.n = 1 << int(t_fit.item())
    n = a.n
    bits = a.index if a.index else (32 if world * n <= (1 << 31) else 64)
    if world * n > 0xFFFFFFFE:
        bits = 64
    w = bits // 8
    box = [psac_amd.unique_id() if rank == 0 else None]
    dist.broadcast_object_list(box, src=0)
    mg = psac_amd.MultiContext.for_rank(rank, world, local_rank, box[0])      # raises if the communicator cannot be built
    if mg.transport != "rccl" or mg.nranks != world:
        raise RuntimeError("bench.py --gpus %d needs the RCCL communicator over %d ranks, got transport %s over %d"
                           % (world, world, mg.transport, mg.nranks))
    lib = mg._lib
    ctx = mg.rank_ctx(0)

    def alloc(nbytes):
        p = C.c_void_p()
        rc = lib.psacx_dev_alloc(ctx, C.byref(p), nbytes)
        if rc != 0:
            raise RuntimeError("device allocation of %d bytes failed" % nbytes)
        return p.value

    d_text = alloc(n)
    rc = lib.psacx_synth_text_dev(ctx, C.c_void_p(d_text), n, rank * n, KIND_ID[a.alphabet], a.seed, 1024)
    assert rc == 0
    slack = n // 8 + 256               # lets the reduced-memory layout use the result arrays as record arrays (psacx.h)
    d_sa, d_isa, d_lcp = alloc((n + slack) * w), alloc((n + slack) * w), alloc((n + slack) * w)
    mg.configure(output_slack=slack)

    def step():
        return mg.construct_device([d_text], [n], [d_sa], [d_isa], None if a.no_lcp else [d_lcp], bits)

    def barrier():
        dist.barrier()
        torch.cuda.synchronize()
        lib.psacx_sync(ctx)

    for _ in range(a.warmup):
        step()
    barrier()
    lib.psacx_profile(ctx, 1)
    t0 = time.perf_counter()
    for _ in range(a.steps):
        st, sent, nex, nga = step()
    barrier()
    dt = time.perf_counter() - t0
    t = torch.tensor([dt], dtype=torch.float64, device="cuda")
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    dt = float(t.item())
    s = psac_amd._lib.Stats()
    lib.psacx_get_stats(ctx, C.byref(s))
    # what every rank saw in its last step: the wire calls it issued, the time its exchanges held its second stream and the
    # host wall time of its phases
    peak_r, reduced_r, slab_rounds_r = mg.memory()
    mine = {"rank": rank, "wire": mg.wire(), "phases_ms": dict((k, round(v, 3)) for k, v in mg.phases()), "payload_bytes_sent": sent,
            "ranks_seen_by_rccl": mg.nranks, "transport": mg.transport, "forms": mg.last_form(),
            "engine_words_per_char_at_peak": round(peak_r[0] / float(n * w), 2),
            "wire_piece_bytes": int(os.environ.get("PSACX_MULTI_WIRE_PIECE", str(1 << 28)))}
    per_rank = [None] * world
    dist.all_gather_object(per_rank, mine)
    if rank == 0:
        out = report(a, world, n, bits, dt, [s.ms_sort_scatter, s.ms_sort_scatter3, s.ms_sort_scatter2], list(s.scatter_bytes),
                     list(s.scatter_launches), None, int(st.k), int(st.bits_per_char), int(st.n_rounds),
                     "block-partitioned text, 1 rank per GPU, C++ host over RCCL (grouped ncclSend/ncclRecv: sort shuffle, "
                     "ISA scatter, B2 fetch, range minima) on a second stream per GPU", int(s.onew_passes), list(s.scatter_records))
        out["exchange"] = {"payload_bytes_sent_by_rank0_per_step": sent, "all_to_all_exchanges_per_step": nex,
                           "scalar_all_gathers_per_step": nga, "uses_rccl": mg.uses_rccl, "transport": mg.transport,
                           "ranks_seen_by_rccl": [p["ranks_seen_by_rccl"] for p in per_rank],
                           "wire_piece_bytes": mine["wire_piece_bytes"], "forms_per_rank": [p["forms"] for p in per_rank],
                           "exchange_ms_on_second_stream_per_rank": [p["wire"]["exchange_ms"][0] for p in per_rank],
                           "nccl_calls_last_step_rank0": {k: mine["wire"][k] for k in ("sends", "recvs", "allgathers")}}
        out["phase_ms_last_step_per_rank"] = [p["phases_ms"] for p in per_rank]
        peak, reduced, slab_rounds = mg.memory()
        out["config"]["layout"] = {"reduced_memory": reduced, "refinement_rounds_in_slabs": slab_rounds,
                                   "engine_words_per_char_at_peak": round(peak[0] / float(n * w), 2),
                                   "engine_words_per_char_at_peak_per_rank": [p["engine_words_per_char_at_peak"] for p in per_rank],
                                   "result_arrays_words_per_char": round(3.0 * (n + slack) / n, 3)}
        # for a like-for-like scaling figure: the one-GPU engine on rank 0's block alone (same size, same index width, the same
        # buffers), timed after the measured region; weak_scaling_efficiency = value / (N x that rate)
        try:
            lib.psacx_trim(ctx)              # the multi-GPU engine's cached blocks go back to the device first
            one = psac_amd.Context(local_rank)
            sa1 = psac_amd.SuffixArray(index_bits=bits, lcp=not a.no_lcp, ctx=one)
            sa1.construct_device(d_text, n, d_sa, d_isa, None if a.no_lcp else d_lcp)
            t1 = time.perf_counter()
            for _ in range(3):
                sa1.construct_device(d_text, n, d_sa, d_isa, None if a.no_lcp else d_lcp)
            t1 = (time.perf_counter() - t1) / 3
            one_rate = n / t1 / 1e6
            out["config"]["one_gpu_engine_same_block"] = {"ms": round(t1 * 1e3, 3), "MChars_per_s": round(one_rate, 1)}
            out["weak_scaling_efficiency"] = round(out["value"] / (world * one_rate), 4)
            one.close()
        except Exception as e:          # never let the side measurement break the bench line
            out["config"]["one_gpu_engine_same_block"] = {"error": str(e)[:200]}
            out["weak_scaling_efficiency"] = None
    for p in (d_text, d_sa, d_isa, d_lcp):
        lib.psacx_dev_free(ctx, C.c_void_p(p))
    mg.close()
    dist.destroy_process_group()
    sys.stdout.flush()
    try:
        C.CDLL(None).fflush(None)           # what the libraries left in the C buffer goes to stderr too
    except Exception:
        pass
    os.dup2(real_stdout, 1)
    os.close(real_stdout)
    if rank == 0:
        print(json.dumps(out))
        sys.stdout.flush()


def side_workloads(a, ctx, sa64, d_text, d_sa, d_isa, d_lcp):
    """The other single-GPU configurations of BASELINE.json and the scatter form the north-star names literally, timed in
    the same run AFTER the timed region (buffers of the default workload reused; the DNA text is restored at the end)."""
    import ctypes as C
    import psac_amd
    lib = ctx._lib
    res = {}

    def run(tag, kind, n, bits, steps, options=None, seed=None, period=1024, recurrence_check=False):
        # recurrence_check: verified by the distributed checker with one rank (every LCP entry through its recurrence) -- the device
        # checker compares characters, sum(LCP) of them: hours on a long repeat
        # options: psacx_configure options of this workload (when the wrappers take their options from PSACX_* variables -- the test suite's
        # debug shim -- the variable of the same name is set as well)
        old = {}
        try:
            for k_, v_ in (options or {}).items():
                if psac_amd._lib.ENV_KNOBS:
                    old["PSACX_" + k_.upper()] = os.environ.get("PSACX_" + k_.upper()); os.environ["PSACX_" + k_.upper()] = str(v_)
            ctx.configure(**(options or {}))
            ctx.check(lib.psacx_synth_text_dev(ctx.handle, C.c_void_p(d_text), n, 0, KIND_ID[kind], a.seed if seed is None else seed, period))
            s_ = sa64 if bits == 64 else psac_amd.SuffixArray(index_bits=32, lcp=True, ctx=ctx)
            s_.construct_device(d_text, n, d_sa, d_isa, d_lcp)
            ctx.check(lib.psacx_sync(ctx.handle))
            t0 = time.perf_counter()
            ms = [0.0, 0.0, 0.0]; by = [0, 0, 0]; la = [0, 0, 0]
            for _ in range(steps):
                st = s_.construct_device(d_text, n, d_sa, d_isa, d_lcp, profile=True)
                ms[0] += st.ms_sort_scatter; ms[1] += st.ms_sort_scatter3; ms[2] += st.ms_sort_scatter2
                for q in (0, 1, 2):
                    by[q] += st.scatter_bytes[q]; la[q] += st.scatter_launches[q]
            ctx.check(lib.psacx_sync(ctx.handle))
            dt = (time.perf_counter() - t0) / steps
            q = max((0, 1, 2), key=lambda j: by[j])
            gbs = by[q] / (ms[q] * 1e-3) / 1e9 if ms[q] > 0 else 0.0
            if recurrence_check:
                ctx.check(lib.psacx_trim(ctx.handle))          # (the workspace of the construction goes back to the device: the checker brings its own)
                mgc = psac_amd.MultiContext([ctx.device])
                try:
                    err = mgc.check_device([d_text], [n], [d_sa], [d_isa], [d_lcp], bits)
                finally:
                    mgc.close()
            else:
                err = psac_amd.check_device(ctx, d_text, n, d_sa, d_isa, d_lcp, bits)
            res[tag] = {"n": n, "index_bits": bits, "ms_per_construction": round(dt * 1e3, 3), "MChars_per_s": round(n / dt / 1e6, 1),
                        "timing": "host wall clock over %d constructions after one warm-up call, text and results resident in HBM" % steps,
                        "rounds": int(st.n_rounds), "verified": list(err) == [0, 0, 0, 0],
                        "scatter_pass": {"form": ("look-back", "three-word (B1,B2,idx)", "two-word (B1,idx)")[q],
                                         "launches_per_construction": la[q] // steps, "avg_launch_ms": round(ms[q] / max(la[q], 1), 4),
                                         "bytes_per_record_per_pass": round(by[q] / max(la[q], 1) / float(n), 2),
                                         "achieved_GBs": round(gbs, 1), "frac_of_8TBs": round(gbs / HBM_PEAK_GBS, 4)}}
        except Exception as e:              # a side measurement never breaks the bench line
            res[tag] = {"error": str(e)[:200]}
        finally:
            ctx.configure(reset=0)
            for k_, v_ in old.items():
                if v_ is None:
                    os.environ.pop(k_, None)
                else:
                    os.environ[k_] = v_

    run("configs[2]: 4096 MiB random ASCII (sigma 128), uint64", "ascii128", 1 << 32, 64, 3)
    run("configs[1]: 256 MiB random DNA, uint32", "dna", 1 << 28, 32, 5)
    # ANSV over the LCP array that run left in HBM (the pass psac -t makes, suffix_tree.hpp:62: left furthest_eq, right nearest_sm)
    res["ansv psac -t pair (furthest_eq, nearest_sm), LCP of 2^28 DNA, uint32"] = ansv_leg(ctx, d_lcp, 1 << 28, 32, d_sa, d_isa)
    # texts that are not random (the reference's own benchmark input is a genome, pbs_run.sh:36): the twin of configs[4] and repeated
    # reads with mutations; host wall time per construction after a warm-up call
    run("configs[4] twin (/256): 128 MiB period-1024 tandem repeat of DNA, uint64", "tandem", 1 << 27, 64, 2, seed=3, recurrence_check=True)
    run("repeated reads with mutations: 1024 MiB (period 65536, one substitution in 200), uint64", "mutated", 1 << 30, 64, 2, seed=7, period=1 << 16,
        recurrence_check=True)
    # configs[4]'s per-GPU block on the one-GPU engine: 2^32 characters of the period-1024 tandem repeat (28 rounds, the reduced-memory
    # layout with its rounds in slabs; rank requests through partition levels and the heavy / light split, psac_amd/csrc/heavy_keys.hpp)
    run("configs[4] block on one GPU: 4096 MiB period-1024 tandem repeat of DNA, uint64", "tandem", 1 << 32, 64, 1, seed=a.seed, recurrence_check=True)
    # the (B1,B2,idx) records of idxsort.hpp:58-62 through every digit of both words (psacx_configure: PSACX_OPT_ONE_STAGE switches the
    # two-stage first round off): 6w = 48 bytes per record and pass, SURVEY 8(d)'s per-unit figure
    run("three-word scatter form: 2048 MiB random DNA, uint64, one-stage first round", "dna", 1 << 31, 64, 2, {"one_stage": 1})
    # the multi-GPU engine on ONE rank with every message through RCCL (no 8-GPU node is needed to time the rank's own work: partition,
    # self-send of the shuffle, bucket passes, ties, rebucket, slice inversion): 2^31 characters of random DNA, uint64
    res["multi-GPU engine, 1 rank, wire forced: 2048 MiB random DNA, uint64"] = multi_engine_leg(a, ctx, d_text, d_sa, d_isa, d_lcp)
    ctx.check(lib.psacx_synth_text_dev(ctx.handle, C.c_void_p(d_text), 1 << 32, 0, KIND_ID[a.alphabet], a.seed, 1024))
    return res


def multi_engine_leg(a, ctx, d_text, d_sa, d_isa, d_lcp, lg=31):
    """psacx_multi_construct_dev_u64 on one rank made with PSACX_MULTI_FORCE_WIRE: the rank sends its whole shuffle to itself through
    ncclSend / ncclRecv and gathers its scalars through ncclAllGather, so the line carries a driver-timed figure of the rank overhead
    of the multi-GPU engine although the box has one GPU.  The result arrays of the default workload (2^32 entries) are reused: they hold
    the block and the slack the reduced-memory layout asks for."""
    import ctypes as C
    import psac_amd
    lib = ctx._lib
    # RCCL prints a version banner on the C stdout of the process: while it may, file descriptor 1 is the process's stderr (as in
    # main_distributed), so that the JSON line stays the only thing on the real stdout
    sys.stdout.flush()
    real_stdout = os.dup(1)
    os.dup2(2, 1)
    try:
        return _multi_engine_leg(a, ctx, lib, d_text, d_sa, d_isa, d_lcp, lg)
    finally:
        try:
            C.CDLL(None).fflush(None)
        except Exception:
            pass
        os.dup2(real_stdout, 1)
        os.close(real_stdout)


def _multi_engine_leg(a, ctx, lib, d_text, d_sa, d_isa, d_lcp, lg):
    import ctypes as C
    import psac_amd
    try:
        n = 1 << lg
        ctx.check(lib.psacx_trim(ctx.handle))
        ctx.check(lib.psacx_synth_text_dev(ctx.handle, C.c_void_p(d_text), n, 0, KIND_ID["dna"], a.seed, 1024))
        mg = psac_amd.MultiContext([ctx.device], force_wire=True)
        try:
            slack = n // 8 + 256
            mg.configure(output_slack=slack)
            mg.construct_device([d_text], [n], [d_sa], [d_isa], [d_lcp], 64)
            times = []
            for _ in range(3):
                t0 = time.perf_counter()
                st, sent, nex, nga = mg.construct_device([d_text], [n], [d_sa], [d_isa], [d_lcp], 64)
                times.append(time.perf_counter() - t0)
            dt = sum(times) / len(times)
            phases, form, wire = mg.phases(), mg.last_form(), mg.wire()
            peak, reduced, slab_rounds = mg.memory()
            err = mg.check_device([d_text], [n], [d_sa], [d_isa], [d_lcp], 64)
            return {"n": n, "index_bits": 64, "ms_per_construction": round(dt * 1e3, 3), "best_ms": round(min(times) * 1e3, 3), "MChars_per_s": round(n / dt / 1e6, 1),
                    "timing": "host wall clock over 3 constructions after one warm-up call, text and results resident in HBM",
                    "ranks_seen_by_rccl": mg.nranks, "transport": mg.transport, "uses_rccl": mg.uses_rccl, "rounds": int(st.n_rounds),
                    "payload_bytes_sent_to_other_ranks": int(sent), "nccl_calls_last_construction": {k: wire[k] for k in ("sends", "recvs", "allgathers")},
                    "exchange_ms_on_second_stream": wire["exchange_ms"][0], "forms": form, "layout_reduced": bool(reduced),
                    "engine_words_per_char_at_peak": round(peak[0] / float(n * 8), 2),
                    "phases_ms_last_construction": dict((k.strip(), round(v, 3)) for k, v in phases), "verified": list(err) == [0, 0, 0, 0]}
        finally:
            mg.close()
    except Exception as e:              # a side measurement never breaks the bench line
        return {"error": str(e)[:300]}


def ansv_leg(ctx, d_in, n, bits, d_l, d_r):
    """psacx_ansv_dev_* on an array resident in HBM (left furthest_eq, right nearest_sm), timed over three calls after a warm-up;
    a window of 2^22 elements in the middle is compared with the oracle's ansv (ansv.hpp:48-65 restated) wherever the oracle's
    answer lies inside the window.  d_l, d_r: room for n uint64 results each (buffers of the workload before it, idle by now)."""
    import numpy as np
    import psac_amd
    import oracle_lib as O
    try:
        w = bits // 8
        NONSV = (1 << 64) - 1
        lt, rt = psac_amd.FURTHEST_EQ, psac_amd.NEAREST_SM
        psac_amd.ansv_device(ctx, d_in, n, d_l, d_r, bits, lt, rt, NONSV)
        ctx.check(ctx._lib.psacx_sync(ctx.handle))
        t0 = time.perf_counter()
        for _ in range(3):
            psac_amd.ansv_device(ctx, d_in, n, d_l, d_r, bits, lt, rt, NONSV)
        ctx.check(ctx._lib.psacx_sync(ctx.handle))
        dt = (time.perf_counter() - t0) / 3
        # the window: values and both results back to the host
        W = min(n, 1 << 22); off = (n - W) // 2
        v = np.empty(W, np.uint32 if bits == 32 else np.uint64); gl = np.empty(W, np.uint64); gr = np.empty(W, np.uint64)
        ctx.d2h(v, d_in + off * w); ctx.d2h(gl, d_l + off * 8); ctx.d2h(gr, d_r + off * 8)
        sm_l = O.ansv(v, True, 0, NONSV); fe_l = O.ansv(v, True, 2, NONSV); sm_r = O.ansv(v, False, 0, NONSV)
        # left (furthest_eq: from the nearest value <= in[i] on through the values equal to THAT one while nothing smaller lies between,
        # ansv_common.hpp:20-22): the window's answer is the global one where something smaller than the answer's value precedes it
        # inside the window, so that the walk ends there
        has = fe_l != np.uint64(NONSV)
        okl = np.zeros(W, bool)
        okl[has] = sm_l[fe_l[has].astype(np.int64)] != np.uint64(NONSV)
        want_l = fe_l + np.uint64(off)
        okr = sm_r != np.uint64(NONSV)
        bad = int(np.count_nonzero(gl[okl] != want_l[okl])) + int(np.count_nonzero(gr[okr] != (sm_r[okr] + np.uint64(off))))
        gbs = n * (w + 16) / dt / 1e9
        return {"n": n, "index_bits": bits, "ms": round(dt * 1e3, 3), "G_elements_per_s": round(n / dt / 1e9, 2),
                "algorithmic_bytes_per_element": w + 16, "achieved_GBs": round(gbs, 1), "frac_of_8TBs": round(gbs / HBM_PEAK_GBS, 4),
                "timing": "host wall clock over 3 calls after one warm-up call, input and results resident in HBM",
                "verified": bad == 0, "checked": "%d left and %d right answers of a 2^22-element window against the oracle" % (int(okl.sum()), int(okr.sum()))}
    except Exception as e:              # a side measurement never breaks the bench line
        return {"error": str(e)[:200]}


def mem_available_bytes():
    try:
        for line in open("/proc/meminfo"):
            if line.startswith("MemAvailable:"):
                return int(line.split()[1]) * 1024
    except OSError:
        pass
    return 0


KIND_ID = {"dna": 0, "ascii128": 1, "tandem": 2, "mutated": 3}


def main():
    a = parse()
    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world > 1 or os.environ.get("PSACX_BENCH_FORCE_DIST"):
        return main_distributed(a, rank, world, local_rank)
    import ctypes as C
    import numpy as np
    import torch
    import psac_amd

    n = a.n
    bits = a.index if a.index else (32 if n <= (1 << 31) else 64)
    w = bits // 8
    ctx = psac_amd.Context(local_rank)
    lib = ctx._lib
    # the text is generated where it is used (psacx_synth_text_dev: the splitmix64 streams of SURVEY 8(d))
    d_text = ctx.alloc(n)
    ctx.check(lib.psacx_synth_text_dev(ctx.handle, C.c_void_p(d_text), n, 0, KIND_ID[a.alphabet], a.seed + rank, 1024))
    d_sa = ctx.alloc(n * w); d_isa = ctx.alloc(n * w); d_lcp = ctx.alloc(n * w)
    sa = psac_amd.SuffixArray(index_bits=bits, lcp=not a.no_lcp, ctx=ctx)

    def step(profile):
        return sa.construct_device(d_text, n, d_sa, d_isa, None if a.no_lcp else d_lcp, profile=profile)

    def barrier():
        torch.cuda.synchronize()
        ctx.check(ctx._lib.psacx_sync(ctx.handle))

    for _ in range(a.warmup):
        step(False)
    barrier()
    # dominant kernel: the scatter kernel of a radix pass.  Large sorts use
    # radix_scatter3_kernel (index 1, or 2 for two-word records), small ones radix_scatter_kernel (index 0); the
    # roofline is quoted on whichever moved more bytes in the timed region.
    scat_ms = [0.0, 0.0, 0.0]; scat_bytes = [0, 0, 0]; scat_launches = [0, 0, 0]; onew = 0
    t0 = time.perf_counter()
    for _ in range(a.steps):
        s = step(True)
        onew += s.onew_passes
        scat_ms[0] += s.ms_sort_scatter; scat_ms[1] += s.ms_sort_scatter3; scat_ms[2] += s.ms_sort_scatter2
        for q in (0, 1, 2):
            scat_bytes[q] += s.scatter_bytes[q]; scat_launches[q] += s.scatter_launches[q]
    barrier()
    dt = time.perf_counter() - t0

    phases = {"total": round(s.ms_total, 3), "alphabet": round(s.ms_alphabet, 3), "kmer": round(s.ms_kmer, 3),
              "sort_hist": round(s.ms_sort_hist, 3), "sort_scatter": round(s.ms_sort_scatter + s.ms_sort_scatter3 + s.ms_sort_scatter2, 3),
              "sort_tile_hist": round(s.ms_sort_tilehist, 3), "rebucket": round(s.ms_rebucket, 3),
              "isa_scatter": round(s.ms_isa_scatter, 3), "gather": round(s.ms_gather, 3), "compact": round(s.ms_compact, 3),
              "rmq_build": round(s.ms_rmq_build, 3)}
    out = report(a, 1, n, bits, dt, scat_ms, scat_bytes, scat_launches, phases, int(s.k), int(s.bits_per_char),
                 int(s.n_rounds), "1 process per GPU", onew)
    out["config"]["workspace_GiB"] = round(s.workspace_bytes / 2.0 ** 30, 1)
    # the result of the last timed step, verified where it lies (psacx_check_dev_*: SA a permutation inverse to ISA,
    # suffix order, every LCP entry against a direct character comparison)
    if not a.no_check:
        t1 = time.perf_counter()
        if a.alphabet == "tandem" and not a.no_lcp:
            # psacx_check_dev_* compares characters (linear in sum(LCP): hours on a long tandem repeat); the multi-GPU
            # engine's checker verifies every LCP entry through the recurrence LCP[i] = 1 + min(LCP[ISA[SA[i-1]+1]+1 ..
            # ISA[SA[i]+1]]) instead (check_suffix_array.hpp:207-267 restated) and works with one rank
            ctx.check(lib.psacx_trim(ctx.handle))
            mg = psac_amd.MultiContext([local_rank])
            err = mg.check_device([d_text], [n], [d_sa], [d_isa], [d_lcp], bits)
            mg.close()
            what = "distributed checker with one rank (SA / ISA by ranks, LCP by its recurrence) over the full result of the last timed step"
        else:
            err = psac_amd.check_device(ctx, d_text, n, d_sa, d_isa, None if a.no_lcp else d_lcp, bits)
            what = "device checker over the full result of the last timed step"
        out["check"] = {"verified": list(err) == [0, 0, 0, 0], "errors": list(err), "seconds": round(time.perf_counter() - t1, 2),
                        "what": what}
    if a.side == "auto" and n == (1 << 32) and bits == 64 and a.alphabet == "dna" and not a.no_lcp:
        out["other_workloads"] = side_workloads(a, ctx, sa, d_text, d_sa, d_isa, d_lcp)
    # SURVEY 8(d) Metric 1 spans what psac brackets (src/psac.cpp:95-121): construct() on host memory, i.e. H2D of
    # the text and D2H of SA / ISA / LCP included.  Reported beside `value`, never as `value`.
    host_bytes = n * (1 + w * (2 if a.no_lcp else 3))
    if a.host_path != "off":
        hn = n
        if a.host_path == "auto" and mem_available_bytes() < 1.3 * host_bytes + (8 << 30):
            hn = 1 << 28
        for p in (d_sa, d_isa, d_lcp):
            ctx.free(p)
        d_sa = d_isa = d_lcp = None
        ctx.check(lib.psacx_trim(ctx.handle))
        text = np.empty(hn, np.uint8)
        ctx.d2h(text, d_text)
        hs = psac_amd.SuffixArray(index_bits=bits, lcp=not a.no_lcp, ctx=ctx)
        hs.construct(text)                       # first call pays for page faults of the result arrays + workspace
        t1 = time.perf_counter()
        hs.local_SA, hs.local_B, hs.local_LCP = hs.construct_into(text, hs.local_SA, hs.local_B, hs.local_LCP)
        ht = time.perf_counter() - t1
        sa_bytes = 1 if hn <= (1 << 8) else 2 if hn <= (1 << 16) else 4 if hn <= (1 << 32) else 8
        narrowed = ("they cross PCIe narrowed to the fewest bytes per entry that hold their largest value (SA / ISA %d of %d bytes at this size, "
                    "LCP as few as 1 on random text)" % (sa_bytes, w)) if sa_bytes < w else "entries of SA / ISA cross PCIe in full (nothing to narrow at this size and width)"
        out["construct_host"] = {"ms": round(ht * 1e3, 1), "MChars_per_s": round(hn / ht / 1e6, 1), "n": hn,
                                 "note": "psacx_construct_u%d on host pointers (SURVEY 8(d) Metric 1): H2D of %d MiB of text, %d MiB of results "
                                         "written into the caller's arrays; %s through a ring of pinned "
                                         "buffers and are widened by host threads; second call on touched pageable memory"
                                         % (bits, hn >> 20, (hn * w * (2 if a.no_lcp else 3)) >> 20, narrowed)}
        # BASELINE's metric in its own sense (SURVEY 8(d) Metric 1 = n / construct() wall time with H2D of the text and D2H of SA / ISA / LCP)
        out["value_metric1"] = out["construct_host"]["MChars_per_s"]
        out["value_metric1_definition"] = ("MChars/s of construct() on host pointers, PCIe copies included (SURVEY 8(d) Metric 1, what psac's own timer "
                                           "spans, src/psac.cpp:95-121), n = %d; `value` is the device-resident rate" % hn)
        del text, hs
    if a.cpu_sample > 0:
        out["cpu_baseline"] = cpu_baseline(a.alphabet, min(a.cpu_sample, n), a.seed, bits)
    print(json.dumps(out))
    for p in (d_text, d_sa, d_isa, d_lcp):
        if p:
            ctx.free(p)
    ctx.close()


if __name__ == "__main__":
    main()
