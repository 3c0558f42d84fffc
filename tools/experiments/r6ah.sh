#!/bin/bash
# host-pointer path, early out against not, alternating processes on one box
cd $GRAFT_REPO_ROOT; export PSACX_ENV_KNOBS=1
O=gpurun_out/r6ah; mkdir -p $O; rm -f $O/host_path.txt
for rep in 1 2; do
  PSACX_NO_EARLY_OUT=1 timeout 300 python tools/host_path_time.py 32 3 2>&1 | grep -v amdgpu | sed 's/^/no early out: /' >> $O/host_path.txt
  timeout 300 python tools/host_path_time.py 32 3 2>&1 | grep -v amdgpu | sed 's/^/early out:    /' >> $O/host_path.txt
done
for rep in 1 2; do
  PSACX_NO_EARLY_OUT=1 timeout 300 python tools/host_path_time.py 28 4 2>&1 | grep -v amdgpu | sed 's/^/no early out: /' >> $O/host_path.txt
  timeout 300 python tools/host_path_time.py 28 4 2>&1 | grep -v amdgpu | sed 's/^/early out:    /' >> $O/host_path.txt
done
cut -c1-260 $O/host_path.txt
