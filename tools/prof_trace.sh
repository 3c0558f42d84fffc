#!/bin/bash
# one rocprofv3 kernel-trace pass over the default bench.py workload: tools/prof_trace.sh <tag>
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/$1
mkdir -p $OUT
ARGS="--steps 3 --warmup 1 --side off --host-path off --no-check --cpu-sample 0"
rocprofv3 --kernel-trace --stats -d /tmp/$1 -o bench -- python $R/bench.py $ARGS > $OUT/trace.log 2>&1
python3 $R/tools/rocpd_summary.py /tmp/$1/bench_results.db > $OUT/trace_summary.txt 2>&1
head -30 $OUT/trace_summary.txt | cut -c1-150
