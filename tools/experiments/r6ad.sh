#!/bin/bash
# ANSV after round 6's changes: the many-tile parity test and the fuzz run (the oracle's furthest_eq is linear now)
cd $GRAFT_REPO_ROOT; export PSACX_ENV_KNOBS=1
O=gpurun_out/r6ad; mkdir -p $O
timeout 500 python -m pytest tests -m gpu -q -x -k "ansv or suffix_tree" 2>&1 | tail -3 > $O/pytest_ansv.txt
timeout 400 python -u tools/fuzz_ansv.py 240 5 > $O/fuzz_ansv.txt 2>&1
timeout 100 python tools/ansv_time.py 26 64 2>&1 | grep ANSV > $O/ansv_time.txt
timeout 100 python tools/ansv_time.py 28 32 2>&1 | grep ANSV >> $O/ansv_time.txt
cat $O/pytest_ansv.txt; tail -2 $O/fuzz_ansv.txt | cut -c1-300; cat $O/ansv_time.txt | cut -c1-200
