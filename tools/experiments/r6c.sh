#!/bin/bash
cd $GRAFT_REPO_ROOT
PSACX_ISA_UPDATE=levels python tools/ab_side.py 3 30 65536 2 check 2>&1 | grep -a "^kind"
PSACX_ISA_UPDATE=stores python tools/ab_side.py 3 30 65536 2 2>&1 | grep -a "^kind"
PSACX_ISA_UPDATE=levels python tools/ab_side.py 2 27 1024 2 check 2>&1 | grep -a "^kind"
PSACX_ISA_UPDATE=levels python tools/ab_side.py 2 30 1024 1 2>&1 | grep -a "^kind"
PSACX_ISA_UPDATE=stores python tools/ab_side.py 2 30 1024 1 2>&1 | grep -a "^kind"
