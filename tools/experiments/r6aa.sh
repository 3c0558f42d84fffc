#!/bin/bash
# round 6, after the evidence job: (a) the split-round cleanup of rebucket_refine_kernel, (b) furthest_eq answers beyond the edge looked up once per tile
cd $GRAFT_REPO_ROOT; export PSACX_ENV_KNOBS=1
O=gpurun_out/r6aa; mkdir -p $O
timeout 600 python -m pytest tests -m gpu -q -x -k "ansv or suffix_tree or long_bucket or tandem" 2>&1 | tail -5 > $O/pytest_sel.txt
timeout 200 python tools/ansv_time.py 28 32 > $O/ansv_2p28_u32.txt 2>&1
timeout 200 python tools/ansv_time.py 26 64 > $O/ansv_2p26_u64.txt 2>&1
timeout 200 python tools/fuzz_long.py 60 21 > $O/fuzz_long.txt 2>&1
timeout 200 python tools/ab_side.py 2 30 1024 1 check > $O/tandem_2p30.txt 2>&1
tail -3 $O/*.txt | cut -c1-250
