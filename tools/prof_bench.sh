#!/bin/bash
# rocprofv3 passes over the default bench.py workload (4 GiB random DNA, uint64): kernel trace, then FETCH_SIZE and WRITE_SIZE in
# their own runs (MI355X_MICROARCH.md: the two do not fit one pass).  Summaries: tools/rocpd_summary.py.
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/prof_bench
mkdir -p $OUT
ARGS="--steps 3 --warmup 1 --side off --host-path off --no-check --cpu-sample 0"
rocprofv3 --kernel-trace --stats -d $OUT/trace -o bench -- python $R/bench.py $ARGS > $OUT/trace.log 2>&1
rocprofv3 --pmc FETCH_SIZE -d $OUT/fetch -o bench -- python $R/bench.py $ARGS > $OUT/fetch.log 2>&1
rocprofv3 --pmc WRITE_SIZE -d $OUT/write -o bench -- python $R/bench.py $ARGS > $OUT/write.log 2>&1
for f in trace fetch write; do python3 $R/tools/rocpd_summary.py $OUT/$f/bench_results.db > $OUT/${f}_summary.txt 2>&1; done
head -14 $OUT/trace_summary.txt
