#!/bin/bash
# rebucket_first_kernel with single output streams left out (tools/experiments/rb_ablate.sh): what does each stream cost?
cd $GRAFT_REPO_ROOT; export PSACX_ENV_KNOBS=1
O=gpurun_out/r6aj; mkdir -p $O; rm -f $O/ablate.txt
for f in tools/experiments/ablate/*.so; do
  PSACX_LIB=$PWD/$f timeout 200 python bench.py --steps 3 --warmup 1 --side off --host-path off --no-check --cpu-sample 0 2>/dev/null | python -c "
import sys, json
d = json.loads(sys.stdin.readline())
print('$(basename $f .so)', d['ms_per_step'], {k: v for k, v in d.get('phase_ms_last_step', {}).items()})" >> $O/ablate.txt
done
cat $O/ablate.txt | cut -c1-400
