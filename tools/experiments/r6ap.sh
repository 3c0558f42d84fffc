#!/bin/bash
# flat byte histograms: pure lanes in segments, runs elsewhere -- parity, timings, per-kernel trace of the mutated reads
cd $GRAFT_REPO_ROOT; export PSACX_ENV_KNOBS=1
O=gpurun_out/r6ap; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_parity.py -m gpu -q -x 2>&1 | grep -v "^RCCL\|^HIP version\|^ROCm\|^Hostname\|^Librccl" | tail -3 > $O/pytest_parity.txt
for a in "3 30 65536" "2 27 1024" "2 30 1024" "3 28 65536"; do
  timeout 160 python tools/ab_side.py $a 2 check 2>&1 | grep "^kind" >> $O/ab.txt
done
timeout 200 tools/prof_kind.sh r6ap/mutated30 3 30 65536 1 40 1000 > /dev/null 2>&1
timeout 120 python tools/fuzz.py 45 51 2>&1 | tail -1 > $O/fuzz.txt
timeout 120 python tools/fuzz_long.py 45 52 2>&1 | tail -1 >> $O/fuzz.txt
cat $O/pytest_parity.txt $O/ab.txt $O/fuzz.txt | cut -c1-200
grep -v "^[WE]2026" $O/mutated30/trace_summary.txt | head -12 | cut -c1-120
