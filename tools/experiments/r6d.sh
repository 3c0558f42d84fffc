#!/bin/bash
cd $GRAFT_REPO_ROOT
run() { echo "== $*"; env "$@" 2>&1 | grep -a "^kind\|rror"; }
run PSACX_GATHER=levels python tools/ab_side.py 2 24 1024 2 check
run PSACX_GATHER=levels PSACX_NO_WHOLE=1 python tools/ab_side.py 2 24 1024 2 check
run PSACX_GATHER=levels PSACX_NO_WHOLE=1 PSACX_FORCE_DIET=1 python tools/ab_side.py 2 24 1024 2 check
run PSACX_GATHER=fetch python tools/ab_side.py 2 27 1024 2
run PSACX_GATHER=levels python tools/ab_side.py 2 27 1024 2 check
run PSACX_GATHER=levels PSACX_NO_WHOLE=1 python tools/ab_side.py 2 27 1024 2 check
run PSACX_GATHER=levels PSACX_NO_WHOLE=1 PSACX_ISA_UPDATE=levels python tools/ab_side.py 2 27 1024 2 check
run PSACX_GATHER=fetch python tools/ab_side.py 3 30 65536 2
run PSACX_GATHER=levels python tools/ab_side.py 3 30 65536 2 check
run PSACX_GATHER=levels PSACX_NO_BUCKET_SORT=1 python tools/ab_side.py 3 30 65536 2 check
run PSACX_GATHER=fetch python tools/ab_side.py 2 30 1024 1
run PSACX_GATHER=levels python tools/ab_side.py 2 30 1024 1 check
run PSACX_GATHER=levels PSACX_NO_WHOLE=1 python tools/ab_side.py 2 30 1024 1 check
run PSACX_GATHER=levels PSACX_NO_WHOLE=1 PSACX_ISA_UPDATE=levels python tools/ab_side.py 2 30 1024 1 check
