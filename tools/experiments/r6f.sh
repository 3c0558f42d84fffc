#!/bin/bash
cd $GRAFT_REPO_ROOT
run() { echo "== $*"; env "$@" 2>&1 | grep -a "^kind\|rror\|Traceback\|psacx"; }
run python tools/ab_side.py 2 24 1024 2 check
run PSACX_FORCE_DIET=1 python tools/ab_side.py 2 24 1024 2 check
run python tools/ab_side.py 2 24 64 2 check
run python tools/ab_side.py 2 24 8192 2 check
run python tools/ab_side.py 2 27 1024 2 check
run python tools/ab_side.py 2 30 1024 1 check
tools/prof_kind.sh r6g/tandem30 2 30 1024 1 14 1000
