#!/bin/bash
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r6e
A="--steps 1 --warmup 1 --side off --host-path off --cpu-sample 0 --alphabet tandem"
PSACX_GATHER=levels python bench.py $A > gpurun_out/r6e/tandem_levels.json 2> gpurun_out/r6e/tandem_levels.err
PSACX_GATHER=fetch python bench.py $A --no-check > gpurun_out/r6e/tandem_fetch.json 2> gpurun_out/r6e/tandem_fetch.err
python - <<'PY'
import json
for t in ("levels", "fetch"):
    try:
        d = json.load(open("gpurun_out/r6e/tandem_%s.json" % t))
        print(t, d["ms_per_step"], d.get("check"), d["phase_ms_last_step"])
    except Exception as e:
        print(t, "failed", e)
PY
tools/prof_kind.sh r6e/twin_trace 2 27 1024 1 30 100
