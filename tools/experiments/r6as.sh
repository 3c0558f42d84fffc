#!/bin/bash
# fuzzers over the session's last code
cd $GRAFT_REPO_ROOT; export PSACX_ENV_KNOBS=1
O=gpurun_out/r6as; mkdir -p $O
timeout 200 python tools/fuzz.py 150 101 2>&1 | tail -1 > $O/fuzz.txt
timeout 200 python tools/fuzz_long.py 150 102 2>&1 | tail -1 >> $O/fuzz.txt
timeout 160 python tools/fuzz_multi.py 120 103 2>&1 | tail -1 >> $O/fuzz.txt
timeout 130 python tools/fuzz_dist.py 90 104 2>&1 | tail -1 >> $O/fuzz.txt
timeout 100 python tools/fuzz_ansv.py 60 105 2>&1 | tail -1 >> $O/fuzz.txt
cat $O/fuzz.txt | cut -c1-250
