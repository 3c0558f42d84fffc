#!/bin/bash
cd $GRAFT_REPO_ROOT
tools/prof_kind.sh r6b/mut_trace 3 30 65536 1 5 300
grep -a "^rounds" gpurun_out/r6b/mut_trace/run.log
# the second construction only (the timed one): the timeline from its start
awk '{print}' gpurun_out/r6b/mut_trace/timeline.txt | tail -260
