#!/bin/bash
# rocprofv3 kernel trace of constructions of one synthetic text (psacx_synth_text_dev kinds: 0 DNA, 2 tandem, 3 mutated reads), uint64:
# tools/prof_kind.sh TAG KIND LOG2N PERIOD [REPS]  ->  gpurun_out/TAG/{run.log,trace_summary.txt}
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/$1
mkdir -p $OUT
rocprofv3 --kernel-trace --stats -d /tmp/$1 -o m -- python $R/tools/ab_side.py $2 $3 $4 ${5:-1} > $OUT/run.log 2>&1
python3 $R/tools/rocpd_summary.py /tmp/$1/m_results.db > $OUT/trace_summary.txt 2>&1
python3 $R/tools/rocpd_timeline.py /tmp/$1/m_results.db ${7:-200} > $OUT/timeline.txt 2>&1
grep -a "^kind" $OUT/run.log; head -${6:-40} $OUT/trace_summary.txt | cut -c1-200
rm -rf /tmp/$1
