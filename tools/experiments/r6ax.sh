#!/bin/bash
# three-word scatter passes (98 registers, two workgroups per CU) capped by __launch_bounds__(512, 6) for three: the compiler stops at 80 registers, 49-56 spilled -- a pass of 2^30 records 8.5 -> 10.1 ms, 2^30 mutated reads 654 -> 658 ms, the /256 tandem twin 179.6 -> 187.3: not kept
cd $GRAFT_REPO_ROOT; export PSACX_ENV_KNOBS=1
O=gpurun_out/r6ax; mkdir -p $O
for a in "3 30 65536" "2 27 1024" "2 30 1024" "3 28 65536"; do
  timeout 160 python tools/ab_side.py $a 2 check 2>&1 | grep "^kind" >> $O/ab.txt
done
timeout 200 tools/prof_kind.sh r6ax/mutated30 3 30 65536 1 40 1000 > /dev/null 2>&1
cat $O/ab.txt | cut -c1-150
grep "radix_scatter3" $O/mutated30/trace_summary.txt | cut -c1-120
