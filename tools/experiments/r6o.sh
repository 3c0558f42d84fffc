#!/bin/bash
cd $GRAFT_REPO_ROOT
export PSACX_ENV_KNOBS=1
mkdir -p gpurun_out/r6o
timeout 300 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "long_bucket or options_through" 2>&1 | tail -3
timeout 200 python tools/fuzz_long.py 60 3 2>&1 | grep -a "fuzz_long\|MISMATCH\|rror"
timeout 120 python tools/ab_side.py 2 27 1024 2 check 2>&1 | grep "^kind"
timeout 200 tools/prof_kind.sh r6o/tandem30 2 30 1024 1 12 1000
A="--steps 1 --warmup 1 --side off --host-path off --cpu-sample 0 --alphabet tandem"
timeout 300 python bench.py $A > gpurun_out/r6o/tandem_4g.json 2> gpurun_out/r6o/tandem_4g.err
python - <<'PY'
import json
try:
    d = json.load(open("gpurun_out/r6o/tandem_4g.json"))
    print("4GiB tandem", d["ms_per_step"], d.get("check", {}).get("verified"), d["phase_ms_last_step"])
except Exception as e:
    print("failed", e); print(open("gpurun_out/r6o/tandem_4g.err").read()[-2000:])
PY
