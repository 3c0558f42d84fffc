#!/bin/bash
# PMC passes over tools/ansv_time.py (separate runs per counter set, as MI355X_MICROARCH.md prescribes): where do the
# wave cycles of ansv_tile_kernel go?  Summaries to gpurun_out/$1.
set -u
TAG=${1:-ansv_pmc}
MODE=${2:-one}          # one: (nearest_sm, nearest_sm); t: (furthest_eq, nearest_sm), the pair psac -t uses
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/$TAG
mkdir -p $O
cd /tmp; export TMPDIR=/tmp
for set in "SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY" "SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS SQ_LDS_IDX_ACTIVE SQ_LDS_BANK_CONFLICT" "SQ_INSTS_LDS SQ_INSTS_VALU SQ_INSTS_VMEM_RD SQ_INSTS_SALU" "FETCH_SIZE" "WRITE_SIZE"; do
  name=$(echo $set | tr ' ' '_' | cut -c1-40)
  rocprofv3 --pmc $set -d /tmp/$TAG/$name -o a -- python $R/tools/ansv_time.py 28 32 $MODE > /dev/null 2> $O/$name.err
  python $R/tools/pmc_summary.py /tmp/$TAG/$name/a_results.db | grep -E "kernel|ansv_wave|ansv_seq|pyramid_level" >> $O/pmc.txt
done
rocprofv3 --kernel-trace --stats -d /tmp/$TAG/trace -o a -- python $R/tools/ansv_time.py 28 32 > $O/ansv_time_under_trace.txt 2> $O/trace.err
python $R/tools/prof_summary.py /tmp/$TAG/trace/a_results.db | grep -E "kernel|ansv_wave|ansv_seq|pyramid_level" > $O/kernel_trace_stats.txt
cat $O/pmc.txt $O/kernel_trace_stats.txt
rm -f $O/*.err
