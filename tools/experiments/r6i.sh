#!/bin/bash
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r6i
run() { echo "== $*"; env "$@" 2>&1 | grep -a "^kind\|rror\|Traceback\|psacx"; }
run python tools/ab_side.py 2 24 1024 1 check
run PSACX_FORCE_DIET=1 python tools/ab_side.py 2 24 1024 1 check
run python tools/ab_side.py 2 24 64 1 check
run python tools/ab_side.py 2 27 1024 2 check
tools/prof_kind.sh r6i/tandem30 2 30 1024 1 8 1000
A="--steps 1 --warmup 1 --side off --host-path off --cpu-sample 0 --alphabet tandem"
python bench.py $A > gpurun_out/r6i/tandem_4g.json 2> gpurun_out/r6i/tandem_4g.err
python - <<'PY'
import json
try:
    d = json.load(open("gpurun_out/r6i/tandem_4g.json"))
    print("4GiB tandem", d["ms_per_step"], d.get("check"), d["phase_ms_last_step"])
except Exception as e:
    print("failed", e); print(open("gpurun_out/r6i/tandem_4g.err").read()[-2000:])
PY
