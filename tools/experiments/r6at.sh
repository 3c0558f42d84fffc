#!/bin/bash
# FETCH_SIZE / WRITE_SIZE of the refinement kernels, session's last code (2^30 mutated reads, 2^30 tandem)
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; export PSACX_ENV_KNOBS=1
O=$R/gpurun_out/r6at; mkdir -p $O
for w in "mutated 3 30 65536" "tandem 2 30 1024"; do
  set -- $w
  for ctr in FETCH_SIZE WRITE_SIZE; do
    timeout 200 rocprofv3 --pmc $ctr -d /tmp/r6at/$1_$ctr -o a -- python $R/tools/ab_side.py $2 $3 $4 1 > $O/$1_$ctr.log 2>&1
    python3 $R/tools/rocpd_summary.py /tmp/r6at/$1_$ctr/a_results.db > $O/refine_${ctr}_$1_2p30.txt 2>&1
    rm -rf /tmp/r6at/$1_$ctr
  done
done
head -12 $O/refine_FETCH_SIZE_mutated_2p30.txt | cut -c1-160
