#!/bin/bash
# tools/kernel_pmc.sh <tag> <kernel-name regex> <command...>: SQ / LDS / instruction-mix counters of one kernel, one
# rocprofv3 --pmc pass per counter set (MI355X_MICROARCH.md: separate passes).  Summary to gpurun_out/<tag>/pmc.txt.
set -u
TAG=$1; shift
PAT=$1; shift
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/$TAG
mkdir -p $O
cd /tmp; export TMPDIR=/tmp
: > $O/pmc.txt
for set in "SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY" "SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS SQ_LDS_IDX_ACTIVE SQ_LDS_BANK_CONFLICT" "SQ_INSTS_LDS SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_SMEM" "SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_VALU" "SQ_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_WAVES SQ_ACTIVE_INST_SCA"; do
  name=$(echo $set | tr ' ' '_' | cut -c1-40)
  rocprofv3 --pmc $set -d /tmp/$TAG/$name -o a -- "$@" > /dev/null 2> $O/$name.err
  python $R/tools/pmc_summary.py /tmp/$TAG/$name/a_results.db | grep -E "$PAT" >> $O/pmc.txt
done
cat $O/pmc.txt
rm -f $O/*.err
