#!/bin/bash
# host-pointer path: SA and LCP on the wire from the end of the first round's rebucket kernel (construct.hpp: EarlyOut) -- parity and timing
cd $GRAFT_REPO_ROOT; export PSACX_ENV_KNOBS=1
O=gpurun_out/r6ag; mkdir -p $O
timeout 600 python -m pytest tests -m gpu -q -x -k "host_pointer or host_path or narrow" 2>&1 | grep -v "^RCCL\|^HIP version\|^ROCm\|^Hostname\|^Librccl" | tail -4 > $O/pytest_host.txt
timeout 300 python tools/host_path_time.py 32 3 2>&1 | grep -v amdgpu > $O/host_path.txt
PSACX_NO_EARLY_OUT=1 timeout 300 python tools/host_path_time.py 32 3 2>&1 | grep -v amdgpu | sed 's/^/no early out: /' >> $O/host_path.txt
timeout 300 python tools/host_path_time.py 28 3 2>&1 | grep -v amdgpu >> $O/host_path.txt
cat $O/pytest_host.txt; cut -c1-260 $O/host_path.txt
