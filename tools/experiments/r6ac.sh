#!/bin/bash
# the ANSV kernel after round 6's changes (one carried entry per value, entries replaced by size, an entry per lane when the table is
# brought up to date, furthest_eq answers beyond the edge looked up once per tile): parity, fuzz, counters, timings
cd $GRAFT_REPO_ROOT; export PSACX_ENV_KNOBS=1
O=gpurun_out/r6ac; mkdir -p $O
timeout 300 python -m pytest tests -m gpu -q -x -k "ansv or suffix_tree" 2>&1 | tail -3 > $O/pytest_ansv.txt
timeout 400 python tools/fuzz_ansv.py 240 5 > $O/fuzz_ansv.txt 2>&1
timeout 100 python tools/ansv_time.py 28 32 2>&1 | grep ANSV > $O/ansv_time.txt
timeout 100 python tools/ansv_time.py 26 64 2>&1 | grep ANSV >> $O/ansv_time.txt
timeout 400 bash tools/ansv_pmc.sh r6ac/pmc_t t > /dev/null 2>&1
timeout 400 bash tools/ansv_pmc.sh r6ac/pmc_one one > /dev/null 2>&1
cat $O/pytest_ansv.txt; tail -2 $O/fuzz_ansv.txt | cut -c1-300; cat $O/ansv_time.txt | cut -c1-200
