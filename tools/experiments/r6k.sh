#!/bin/bash
# counters of the first round's kernels on the default workload (what binds key_scatter1w_kernel?)
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r6k
mkdir -p $O
ARGS="--steps 2 --warmup 1 --side off --host-path off --no-check --cpu-sample 0"
PAT="key_scatter1w|top_digit|radix_scatter1w|radix_tile_hist|rebucket_first|partition_packed|window_scatter_packed|tie_resolve|char_hist"
i=0
for set in "TCP_UTCL1_TRANSLATION_MISS_sum TCP_UTCL1_TRANSLATION_HIT_sum TCP_UTCL1_REQUEST_sum" \
           "TCC_EA0_WRREQ_sum TCC_EA0_WRREQ_64B_sum TCC_EA0_WRREQ_STALL_sum TCC_TOO_MANY_EA_WRREQS_STALL_sum" \
           "TCC_EA0_WRREQ_DRAM_CREDIT_STALL_sum TCC_TAG_STALL_sum TCC_IB_STALL_sum" \
           "TCP_PENDING_STALL_CYCLES_sum TCP_TCR_TCP_STALL_CYCLES_sum TA_ADDR_STALLED_BY_TC_CYCLES_sum GRBM_GUI_ACTIVE" \
           "SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_SALU" \
           "TCP_UTCL1_STALL_MULTI_MISS_sum TCP_UTCL1_STALL_UTCL2_REQ_OUT_OF_CREDITS_sum TCP_UTCL1_THRASHING_STALL_sum GRBM_UTCL2_BUSY"; do
  i=$((i+1))
  rocprofv3 --pmc $set -d /tmp/r6k/p$i -o a -- python $R/bench.py $ARGS > $O/p$i.log 2>&1
  python3 $R/tools/rocpd_summary.py /tmp/r6k/p$i/a_results.db > $O/p$i.txt 2>&1
  echo "== $set"; sed -n '/^counters/,$p' $O/p$i.txt | grep -E "$PAT" | cut -c1-150
  rm -rf /tmp/r6k/p$i
done
