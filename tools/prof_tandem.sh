#!/bin/bash
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/prof_tandem
mkdir -p $OUT
ARGS="--steps 1 --warmup 0 --side off --host-path off --no-check --cpu-sample 0 --alphabet tandem"
rocprofv3 --kernel-trace --stats -d $OUT/trace -o bench -- python $R/bench.py $ARGS > $OUT/trace.log 2>&1
python3 $R/tools/rocpd_summary.py $OUT/trace/bench_results.db > $OUT/trace_summary.txt 2>&1
head -50 $OUT/trace_summary.txt
tail -3 $OUT/trace.log
rm -rf $OUT/trace
