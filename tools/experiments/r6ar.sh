#!/bin/bash
# ANSV of the final code (types as template parameters): times per pair and the counters of the psac -t pair
cd $GRAFT_REPO_ROOT; export PSACX_ENV_KNOBS=1
O=gpurun_out/r6ar; mkdir -p $O
(echo "tools/ansv_time.py 28 32; tools/ansv_time.py 26 64 (final code of round 6; host wall clock per call, pyramid build included)"; timeout 120 python tools/ansv_time.py 28 32 2>&1 | grep "^ANSV"; timeout 120 python tools/ansv_time.py 26 64 2>&1 | grep "^ANSV") > $O/ansv_time.txt
timeout 400 bash tools/ansv_pmc.sh r6ar/pmc_t t > /dev/null 2>&1
timeout 100 python tools/fuzz_ansv.py 40 61 2>&1 | tail -1 > $O/fuzz_ansv.txt
cat $O/ansv_time.txt $O/pmc_t/pmc.txt $O/pmc_t/kernel_trace_stats.txt $O/fuzz_ansv.txt | cut -c1-220
