#!/bin/bash
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r6j
mkdir -p $O
rocprofv3 -L > $O/counters_all.txt 2>&1
grep -i -o -E "\b(TCP|TCC|TCA|TA|TD|SQ|GRBM|GL2C|UTCL2|ATC)[A-Z0-9_]*\b" $O/counters_all.txt | sort -u > $O/counter_names.txt
wc -l $O/counter_names.txt
grep -i -E "UTCL|TLB|XNACK|STALL|WRREQ|EA0_WR|ATC|MISS" $O/counter_names.txt | tr '\n' ' '
