#!/bin/bash
# variants of the ANSV kernel side by side (tools/experiments/ansv_ablate.sh builds them)
cd $GRAFT_REPO_ROOT; export PSACX_ENV_KNOBS=1
O=gpurun_out/r6ab; mkdir -p $O; rm -f $O/ablate.txt
for f in tools/experiments/ablate/*.so; do
  for a in "28 32 one"; do
  PSACX_LIB=$PWD/$f timeout 100 python tools/ansv_time.py $a 2>&1 | grep ANSV | sed "s/^/$(basename $f .so): /" >> $O/ablate.txt
  done
done
cut -c1-190 $O/ablate.txt
