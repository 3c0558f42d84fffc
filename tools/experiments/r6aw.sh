#!/bin/bash
# rebucket_refine_kernel (one GPU, LCP) capped at 128 registers: two workgroups per CU (24 / 42 registers spilled) against one at 144 / 152
cd $GRAFT_REPO_ROOT; export PSACX_ENV_KNOBS=1
O=gpurun_out/r6aw; mkdir -p $O
for a in "3 30 65536" "2 27 1024" "2 30 1024" "3 28 65536"; do
  timeout 160 python tools/ab_side.py $a 2 check 2>&1 | grep "^kind" >> $O/ab.txt
done
timeout 200 tools/prof_kind.sh r6aw/mutated30 3 30 65536 1 40 1000 > /dev/null 2>&1
cat $O/ab.txt | cut -c1-150
grep "rebucket_refine" $O/mutated30/trace_summary.txt | cut -c1-100; grep "rebucket_refine" $O/mutated30/timeline.txt | awk '{print $3}' | tr '\n' ' '
