#!/bin/bash
# Runs on the GPU box: kernel-trace stats and the two HBM-traffic PMC passes (separate runs, as
# MI355X_MICROARCH.md prescribes) for the default bench workload; text summaries go to gpurun_out/$1.
set -u
TAG=${1:-prof}
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/$TAG
mkdir -p $O
cd /tmp; export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats -d /tmp/$TAG/trace -o bench -- python $R/bench.py --steps 5 --warmup 1 --cpu-sample 0 --host-path off --no-check > $O/bench_under_trace.json 2> $O/trace.err
python $R/tools/prof_summary.py /tmp/$TAG/trace/bench_results.db > $O/kernel_trace_stats.txt
rocprofv3 --pmc FETCH_SIZE -d /tmp/$TAG/fetch -o bench -- python $R/bench.py --steps 1 --warmup 0 --cpu-sample 0 --host-path off --no-check > /dev/null 2> $O/fetch.err
python $R/tools/pmc_summary.py /tmp/$TAG/fetch/bench_results.db > $O/pmc_fetch_size.txt
rocprofv3 --pmc WRITE_SIZE -d /tmp/$TAG/write -o bench -- python $R/bench.py --steps 1 --warmup 0 --cpu-sample 0 --host-path off --no-check > /dev/null 2> $O/write.err
python $R/tools/pmc_summary.py /tmp/$TAG/write/bench_results.db > $O/pmc_write_size.txt
tail -n 3 $O/*.err | grep -v "^$" | tail -5
rm -f $O/*.err
cat $O/kernel_trace_stats.txt | head -12
grep -E "scatter|counter" $O/pmc_fetch_size.txt $O/pmc_write_size.txt
