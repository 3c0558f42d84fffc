#!/bin/bash
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r6l
for v in "" "PSACX_NO_REGIONS=1"; do
env $v python bench.py --steps 5 --warmup 1 --side off --host-path off --cpu-sample 0 > gpurun_out/r6l/bench.json 2> gpurun_out/r6l/bench.err
python - <<'PY'
import json
try:
    d = json.load(open("gpurun_out/r6l/bench.json"))
    print(d["ms_per_step"], d["check"]["verified"], d["roofline"]["frac"], d["phase_ms_last_step"])
except Exception as e:
    print("failed", e); print(open("gpurun_out/r6l/bench.err").read()[-1500:])
PY
done
python tools/ab_side.py 0 30 0 2 check 2>&1 | grep "^kind"
python tools/ab_side.py 0 29 0 2 check 2>&1 | grep "^kind"
