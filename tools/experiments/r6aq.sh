#!/bin/bash
# round 6, code of the session's end: bench lines, kernel traces of the bench and of the repetitive side workloads, the whole GPU suite
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
export PSACX_ENV_KNOBS=1
O=$R/gpurun_out/r6aq; mkdir -p $O
cd $R
timeout 500 python bench.py > $O/bench_default.json 2> $O/bench_default.err
timeout 200 python bench.py --steps 1 --warmup 1 --side off --host-path off --cpu-sample 0 --alphabet tandem > $O/bench_tandem_4gib.json 2> $O/bench_tandem_4gib.err
cd /tmp
ARGS="--steps 3 --warmup 1 --side off --host-path off --no-check --cpu-sample 0"
timeout 200 rocprofv3 --kernel-trace --stats -d /tmp/r6aq/trace -o bench -- python $R/bench.py $ARGS > $O/trace.log 2>&1
python3 $R/tools/rocpd_summary.py /tmp/r6aq/trace/bench_results.db > $O/bench_trace_4gib_u64.txt 2>&1
rm -rf /tmp/r6aq
cd $R
timeout 200 tools/prof_kind.sh r6aq/twin27 2 27 1024 1 40 1000 > /dev/null 2>&1
timeout 200 tools/prof_kind.sh r6aq/mutated30 3 30 65536 1 40 1000 > /dev/null 2>&1
timeout 200 tools/prof_kind.sh r6aq/tandem30 2 30 1024 1 40 1000 > /dev/null 2>&1
timeout 700 python -m pytest tests -m gpu -q 2>&1 | grep -v "^RCCL\|^HIP version\|^ROCm\|^Hostname\|^Librccl" | tail -6 > $O/pytest_gpu.txt
tail -3 $O/pytest_gpu.txt
python - <<'PY'
import json
for f in ("bench_default", "bench_tandem_4gib"):
    try:
        d = json.loads(open("gpurun_out/r6aq/%s.json" % f).readline())
        print(f, d["ms_per_step"], d["check"]["verified"], d["roofline"]["frac"], d.get("value_metric1"), d.get("construct_host", {}).get("ms"))
        for k, v in d.get("other_workloads", {}).items():
            print("   ", k[:60], v.get("ms_per_construction", v.get("ms")), v.get("frac_of_8TBs"), v.get("verified"))
    except Exception as e:
        print(f, "failed", e)
PY
