#!/bin/bash
# final code of round 6: bench lines and the whole GPU suite
cd $GRAFT_REPO_ROOT; export PSACX_ENV_KNOBS=1
O=gpurun_out/r6ai; mkdir -p $O
timeout 500 python bench.py > $O/bench_default.json 2> $O/bench_default.err
timeout 200 python bench.py --steps 1 --warmup 1 --side off --host-path off --cpu-sample 0 --alphabet tandem > $O/bench_tandem_4gib.json 2> $O/bench_tandem_4gib.err
timeout 900 python -m pytest tests -m gpu -q 2>&1 | grep -v "^RCCL\|^HIP version\|^ROCm\|^Hostname\|^Librccl" | tail -6 > $O/pytest_gpu.txt
cat $O/pytest_gpu.txt
python - <<'PY'
import json
for f in ("bench_default", "bench_tandem_4gib"):
    try:
        d = json.loads(open("gpurun_out/r6ai/%s.json" % f).readline())
        print(f, d["ms_per_step"], d["check"]["verified"], d["roofline"]["frac"], d.get("value_metric1"), d.get("construct_host", {}).get("ms"))
        for k, v in d.get("other_workloads", {}).items():
            print("   ", k[:60], v.get("ms_per_construction", v.get("ms")), v.get("frac_of_8TBs"), v.get("verified"))
    except Exception as e:
        print(f, "failed", e)
PY
