#!/bin/bash
# rocprofv3 passes over the ANSV call on the LCP of 2^28 random DNA characters (tools/ansv_time.py): kernel trace + SQ counters
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/prof_ansv
mkdir -p $OUT
rocprofv3 --kernel-trace --stats -d $OUT/trace -o ansv -- python $R/tools/ansv_time.py 28 32 > $OUT/trace.log 2>&1
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_SALU -d $OUT/pmc1 -o ansv -- python $R/tools/ansv_time.py 28 32 > $OUT/pmc1.log 2>&1
rocprofv3 --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAVES SQ_BUSY_CYCLES SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS -d $OUT/pmc2 -o ansv -- python $R/tools/ansv_time.py 28 32 > $OUT/pmc2.log 2>&1
ls -R $OUT | head -30
