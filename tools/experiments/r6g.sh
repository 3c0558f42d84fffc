#!/bin/bash
cd $GRAFT_REPO_ROOT
tools/prof_kind.sh r6g/tandem30 2 30 1024 1 32 1000
grep -a "^rounds" gpurun_out/r6g/tandem30/run.log | cut -c1-1500
