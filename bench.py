#!/usr/bin/env python3
"""bench.py -- SA+LCP construction throughput of the HIP engine on MI355X.

One "step" = one full suffix-array + inverse-SA + LCP construction of the
synthetic text, text already resident in HBM, results left in HBM.

N = 1 workload (BASELINE.json configs[1]): 256 MiB random DNA (sigma = 4,
splitmix64 seed 1), uint32 indices, SA + LCP.

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
TRAFFIC_PER_LAUNCH = {1: 6553287372, 2: 4184907503}


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--n", type=int, default=1 << 28, help="characters per GPU")
    ap.add_argument("--index", type=int, default=32, choices=(32, 64))
    ap.add_argument("--alphabet", default="dna", choices=("dna", "ascii128", "tandem"))
    ap.add_argument("--seed", type=int, default=1)
    ap.add_argument("--cpu-sample", type=int, default=1 << 28, help="characters for the CPU baseline leg (0 = skip); the default is the whole workload, ~8 s on the GPU box's 256 host threads")
    ap.add_argument("--no-lcp", action="store_true")
    ap.add_argument("--host-path", action="store_true",
                    help="also time psacx_construct_* with host pointers (H2D of the text, D2H of SA/ISA/LCP); reported "
                         "as an extra field, never as `value`")
    return ap.parse_args()


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
    return {"value": round(sample / dt / 1e6, 3), "unit": "MChars/s", "cores": cores, "kind": "port",
            "sample": "%d chars of the same generator (%s, seed %d), SA+LCP, uint%d, %.1f s on %d threads"
                      % (sample, kind, seed, bits, dt, cores)}


def report(a, world, n, bits, dt, scat_ms, scat_bytes, scat_launches, phases, k, l, rounds, parallelism):
    w = bits // 8
    ms_per_step = dt / a.steps * 1e3
    value = world * n * a.steps / dt / 1e6
    dom = max(range(len(scat_bytes)), key=lambda q: scat_bytes[q])
    kname = ("radix_scatter_kernel (one 8-bit digit pass of the (B1,B2,idx) rank-pair sort, look-back form)",
             "radix_scatter3_kernel (one 8-bit digit pass of the (B1,B2,idx) rank-pair sort)",
             "radix_scatter3_kernel<two-word> (one 8-bit digit pass of the first round's (B1,idx) prefix sort)")[dom]
    rec_words = 2 if dom == 2 else 3
    achieved = scat_bytes[dom] / (scat_ms[dom] * 1e-3) / 1e9 if scat_ms[dom] > 0 else 0.0
    out = {
        "metric": "MChars/s SA+LCP build; rank-pair radix-sort HBM GB/s vs peak",
        "value": round(value, 2), "unit": "MChars/s", "n_gpus": world, "steps": a.steps,
        "warmup": a.warmup, "ms_per_step": round(ms_per_step, 3), "higher_is_better": True,
        "scaling": "weak", "vs_baseline": None, "dtype": "u%d" % bits, "data": "synthetic",
        "config": {"workload": "%d MiB random %s per GPU (splitmix64 seed %d + rank), %d MiB in total, uint%d indices, "
                               "SA+%s on %d x MI355X" % (n >> 20, a.alphabet, a.seed, (world * n) >> 20, bits,
                                                         "ISA" if a.no_lcp else "ISA+LCP", world),
                   "n_per_gpu": n, "k": k, "bits_per_char": l, "rounds": rounds, "parallelism": parallelism},
        "roofline": {"bound": "hbm", "kernel": kname,
                     "achieved": round(achieved, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                     "frac": round(achieved / HBM_PEAK_GBS, 4),
                     "avg_launch_ms": round(scat_ms[dom] / max(scat_launches[dom], 1), 4),
                     "launches_per_step": scat_launches[dom] // max(a.steps, 1),
                     "algorithmic_bytes_per_launch": scat_bytes[dom] // max(scat_launches[dom], 1),
                     "bytes_per_record_per_pass": 2 * rec_words * w,
                     "traffic": TRAFFIC_PER_LAUNCH.get(dom) if (world == 1 and n == (1 << 28) and bits == 32) else None},
    }
    if phases:
        out["phase_ms_last_step"] = phases
    return out


def main_distributed(a, rank, world, local_rank):
    """N > 1: the text is block-partitioned over the ranks (one block of --n characters per GPU);
    sort shuffle, SA->ISA scatter, B2 fetch and range-min queries go through RCCL all-to-all."""
    import torch
    import torch.distributed as dist
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    os.environ.setdefault("MASTER_PORT", "29533")
    os.environ.setdefault("RANK", str(rank)); os.environ.setdefault("WORLD_SIZE", str(world))
    torch.cuda.set_device(local_rank)
    dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))
    from psac_amd import dist as D
    from psac_amd.comm import TorchComm
    from psac_amd.dist_ops import HipOps
    n, bits = a.n, a.index
    if world * n > 0xFFFFFFFE:
        bits = 64
    ops = HipOps(bits, local_rank)
    comm = TorchComm()
    text = torch.from_numpy(make_text(a.alphabet, n, a.seed + rank)).cuda()

    def step():
        return D.run(D.construct(comm, ops, text, want_lcp=not a.no_lcp))

    def barrier():
        dist.barrier()
        torch.cuda.synchronize()

    for _ in range(a.warmup):
        step()
    barrier()
    ops.profile(True)
    t0 = time.perf_counter()
    for _ in range(a.steps):
        res = step()
    barrier()
    dt = time.perf_counter() - t0
    t = torch.tensor([dt], dtype=torch.float64, device="cuda")
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    dt = float(t.item())
    s = ops.stats()
    if rank == 0:
        out = report(a, world, n, bits, dt, [s.ms_sort_scatter, s.ms_sort_scatter3, s.ms_sort_scatter2], list(s.scatter_bytes),
                     list(s.scatter_launches), None, res["k"], res["l"], len(res["rounds"]),
                     "block-partitioned text, 1 rank per GPU, RCCL all-to-all (sort shuffle, ISA scatter, B2 fetch)")
        print(json.dumps(out))
    ops.close()
    dist.destroy_process_group()


def main():
    a = parse()
    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world > 1 or os.environ.get("PSACX_BENCH_FORCE_DIST"):
        return main_distributed(a, rank, world, local_rank)
    import numpy as np
    import torch
    import psac_amd

    n = a.n
    bits = a.index
    w = bits // 8
    ctx = psac_amd.Context(local_rank)
    text = make_text(a.alphabet, n, a.seed + rank)
    d_text = ctx.alloc(n)
    ctx.h2d(d_text, text)
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
    # radix_scatter3_kernel (index 1), small ones radix_scatter_kernel (index 0); the
    # roofline is quoted on whichever moved more bytes in the timed region.
    scat_ms = [0.0, 0.0, 0.0]; scat_bytes = [0, 0, 0]; scat_launches = [0, 0, 0]
    t0 = time.perf_counter()
    for _ in range(a.steps):
        s = step(True)
        scat_ms[0] += s.ms_sort_scatter; scat_ms[1] += s.ms_sort_scatter3; scat_ms[2] += s.ms_sort_scatter2
        for q in (0, 1, 2):
            scat_bytes[q] += s.scatter_bytes[q]; scat_launches[q] += s.scatter_launches[q]
    barrier()
    dt = time.perf_counter() - t0

    # sanity on the result of the last step (cheap device->host spot check)
    head = np.empty(4, np.uint32 if bits == 32 else np.uint64)
    ctx.d2h(head, d_lcp if not a.no_lcp else d_sa)

    phases = {"total": round(s.ms_total, 3), "alphabet": round(s.ms_alphabet, 3), "kmer": round(s.ms_kmer, 3),
              "sort_hist": round(s.ms_sort_hist, 3), "sort_scatter": round(s.ms_sort_scatter + s.ms_sort_scatter3 + s.ms_sort_scatter2, 3),
              "sort_tile_hist": round(s.ms_sort_tilehist, 3), "rebucket": round(s.ms_rebucket, 3),
              "isa_scatter": round(s.ms_isa_scatter, 3), "gather": round(s.ms_gather, 3), "compact": round(s.ms_compact, 3),
              "rmq_build": round(s.ms_rmq_build, 3)}
    out = report(a, 1, n, bits, dt, scat_ms, scat_bytes, scat_launches, phases, int(s.k), int(s.bits_per_char),
                 int(s.n_rounds), "1 process per GPU")
    if a.host_path:
        hs = psac_amd.SuffixArray(index_bits=bits, lcp=not a.no_lcp, ctx=ctx)
        hs.construct(text)                       # first call pays for page faults of the result arrays
        t1 = time.perf_counter()
        hs.construct(text)
        ht = time.perf_counter() - t1
        out["host_pointer_path"] = {"ms": round(ht * 1e3, 1), "MChars_per_s": round(n / ht / 1e6, 1),
                                    "note": "pageable host buffers: H2D %d MiB + D2H %d MiB over PCIe, device buffers allocated per call"
                                            % (n >> 20, (n * w * (2 if a.no_lcp else 3)) >> 20)}
    if a.cpu_sample > 0:
        out["cpu_baseline"] = cpu_baseline(a.alphabet, min(a.cpu_sample, n), a.seed, bits)
    print(json.dumps(out))
    for p in (d_text, d_sa, d_isa, d_lcp):
        ctx.free(p)
    ctx.close()


if __name__ == "__main__":
    main()
