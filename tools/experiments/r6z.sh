#!/bin/bash
# the evidence of round 6, final code: everything DESIGN / README quote (copied into profiles/ afterwards)
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
export PSACX_ENV_KNOBS=1
O=$R/gpurun_out/r6z; mkdir -p $O
cd $R
timeout 400 python bench.py > $O/bench_default.json 2> $O/bench_default.err
timeout 200 python bench.py --steps 1 --warmup 1 --side off --host-path off --cpu-sample 0 --alphabet tandem > $O/bench_tandem_4gib.json 2> $O/bench_tandem_4gib.err
cd /tmp
ARGS="--steps 3 --warmup 1 --side off --host-path off --no-check --cpu-sample 0"
timeout 200 rocprofv3 --kernel-trace --stats -d /tmp/r6z/trace -o bench -- python $R/bench.py $ARGS > $O/trace.log 2>&1
timeout 200 rocprofv3 --pmc FETCH_SIZE -d /tmp/r6z/fetch -o bench -- python $R/bench.py $ARGS > $O/fetch.log 2>&1
timeout 200 rocprofv3 --pmc WRITE_SIZE -d /tmp/r6z/write -o bench -- python $R/bench.py $ARGS > $O/write.log 2>&1
for f in trace fetch write; do python3 $R/tools/rocpd_summary.py /tmp/r6z/$f/bench_results.db > $O/bench_${f}_4gib_u64.txt 2>&1; done
rm -rf /tmp/r6z
cd $R
timeout 200 tools/prof_kind.sh r6z/tandem30 2 30 1024 1 40 1000 > /dev/null 2>&1
timeout 200 tools/prof_kind.sh r6z/twin27 2 27 1024 1 40 1000 > /dev/null 2>&1
timeout 200 tools/prof_kind.sh r6z/mutated30 3 30 65536 1 40 1000 > /dev/null 2>&1
cd /tmp
for w in "mutated 3 30 65536" "tandem 2 30 1024"; do
  set -- $w
  for ctr in FETCH_SIZE WRITE_SIZE; do
    timeout 200 rocprofv3 --pmc $ctr -d /tmp/r6z/$1_$ctr -o a -- python $R/tools/ab_side.py $2 $3 $4 1 > $O/$1_$ctr.log 2>&1
    python3 $R/tools/rocpd_summary.py /tmp/r6z/$1_$ctr/a_results.db > $O/refine_${ctr}_$1_2p30.txt 2>&1
    rm -rf /tmp/r6z/$1_$ctr
  done
done
cd $R
timeout 200 python tools/ab_side.py 2 31 1024 1 check > $O/tandem_2p31.txt 2>&1
timeout 200 python tools/ab_side.py 2 29 4096 1 check > $O/tandem_2p29_period4096.txt 2>&1
timeout 60 ./tools/ubench_gather 32 30 > $O/ubench_gather_32_30.txt 2>&1
timeout 60 ./tools/ubench_atomic > $O/ubench_atomic.txt 2>&1
timeout 60 ./tools/ubench_fronts2 > $O/ubench_fronts2.txt 2>&1
timeout 200 python tools/fuzz_long.py 120 11 > $O/fuzz_long.txt 2>&1
timeout 200 python tools/fuzz.py 120 12 > $O/fuzz.txt 2>&1
timeout 600 python -m pytest tests -m gpu -q 2>&1 | tail -6 > $O/pytest_gpu.txt
tail -3 $O/pytest_gpu.txt $O/fuzz_long.txt $O/fuzz.txt
python - <<'PY'
import json
for f in ("bench_default", "bench_tandem_4gib"):
    try:
        d = json.loads(open("gpurun_out/r6z/%s.json" % f).readline())
        print(f, d["ms_per_step"], d["check"]["verified"], d["roofline"]["frac"], d.get("value_metric1"))
        for k, v in d.get("other_workloads", {}).items():
            print("   ", k[:60], v.get("ms_per_construction", v.get("ms")), v.get("verified"))
    except Exception as e:
        print(f, "failed", e)
PY
