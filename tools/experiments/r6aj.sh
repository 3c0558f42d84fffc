#!/bin/bash
# variants of the first-round kernels (tools/experiments/rb_ablate.sh) through bench.py, verified unless parts are left out
cd $GRAFT_REPO_ROOT; export PSACX_ENV_KNOBS=1
O=gpurun_out/r6aj; mkdir -p $O; rm -f $O/ablate.txt
for f in tools/experiments/ablate/*.so; do
  PSACX_LIB=$PWD/$f timeout 200 python bench.py --steps 3 --warmup 1 --side off --host-path off --cpu-sample 0 2>/dev/null | python tools/experiments/bench_phases.py $(basename $f .so) >> $O/ablate.txt
done
cat $O/ablate.txt | cut -c1-400
