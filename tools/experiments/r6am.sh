#!/bin/bash
# digit bytes between the three-kernel passes of the refinement sorts (dispatch_pass3): parity, then A/B against PSACX_NO_DIGIT_BYTES=1
cd $GRAFT_REPO_ROOT; export PSACX_ENV_KNOBS=1
O=gpurun_out/r6am; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_parity.py -m gpu -q -x 2>&1 | grep -v "^RCCL\|^HIP version\|^ROCm\|^Hostname\|^Librccl" | tail -3 > $O/pytest_parity.txt
for a in "3 30 65536" "2 27 1024" "2 30 1024" "3 28 65536"; do
  timeout 120 python tools/ab_side.py $a 2 check 2>&1 | grep "^kind" | sed 's/^/bytes  /' >> $O/ab.txt
  PSACX_NO_DIGIT_BYTES=1 timeout 120 python tools/ab_side.py $a 2 2>&1 | grep "^kind" | sed 's/^/records /' >> $O/ab.txt
done
timeout 150 python tools/fuzz.py 60 41 2>&1 | tail -1 > $O/fuzz.txt
timeout 150 python tools/fuzz_long.py 60 42 2>&1 | tail -1 >> $O/fuzz.txt
cat $O/pytest_parity.txt $O/ab.txt $O/fuzz.txt | cut -c1-200
