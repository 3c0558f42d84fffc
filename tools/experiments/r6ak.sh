#!/bin/bash
# rebucket_refine_kernel with its range-minimum walk inlined once instead of per record (78 -> 25 KB of code): parity and timing
cd $GRAFT_REPO_ROOT; export PSACX_ENV_KNOBS=1
O=gpurun_out/r6ak; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_parity.py -m gpu -q -x 2>&1 | grep -v "^RCCL\|^HIP version\|^ROCm\|^Hostname\|^Librccl" | tail -3 > $O/pytest_parity.txt
for a in "3 30 65536" "3 28 65536" "2 27 1024" "2 30 1024" "0 30 0"; do timeout 120 python tools/ab_side.py $a 1 check 2>&1 | grep "^kind" >> $O/ab.txt; done
timeout 200 python tools/fuzz.py 90 31 2>&1 | tail -1 > $O/fuzz.txt
timeout 200 python tools/fuzz_long.py 60 32 2>&1 | tail -1 >> $O/fuzz.txt
cat $O/pytest_parity.txt $O/ab.txt $O/fuzz.txt | cut -c1-200
