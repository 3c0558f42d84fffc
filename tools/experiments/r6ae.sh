#!/bin/bash
# final ANSV kernel of round 6: whole GPU suite, fuzz, bench line, counters of the psac -t pair and the nearest pair
cd $GRAFT_REPO_ROOT; export PSACX_ENV_KNOBS=1
O=gpurun_out/r6ae; mkdir -p $O
timeout 900 python -m pytest tests -m gpu -q 2>&1 | tail -4 > $O/pytest_gpu.txt
timeout 300 python -u tools/fuzz_ansv.py 180 6 > $O/fuzz_ansv.txt 2>&1
timeout 400 python bench.py > $O/bench_default.json 2> $O/bench_default.err
timeout 400 bash tools/ansv_pmc.sh r6ae/pmc_t t > /dev/null 2>&1
timeout 400 bash tools/ansv_pmc.sh r6ae/pmc_one one > /dev/null 2>&1
timeout 100 python tools/ansv_time.py 28 32 2>&1 | grep ANSV > $O/ansv_time.txt
timeout 100 python tools/ansv_time.py 26 64 2>&1 | grep ANSV >> $O/ansv_time.txt
cat $O/pytest_gpu.txt; tail -1 $O/fuzz_ansv.txt | cut -c1-300; cut -c1-200 $O/ansv_time.txt
python - <<'PY'
import json
d = json.loads(open("gpurun_out/r6ae/bench_default.json").readline())
print(d["ms_per_step"], d["check"]["verified"], d["roofline"]["frac"])
for k, v in d.get("other_workloads", {}).items():
    print("   ", k[:60], v.get("ms_per_construction", v.get("ms")), v.get("frac_of_8TBs"), v.get("verified"))
PY
