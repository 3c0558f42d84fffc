#!/bin/bash
# Variants of libpsacx.so whose first-round kernels leave parts out (RB_ABLATE, sa_kernels.hpp) -- to time the parts; results are wrong.
# usage: rb_ablate.sh tag=-DRB_ABLATE=1 ...   (run here; PSACX_LIB selects a variant on the GPU box)
cd /root/repo/psac_amd/csrc
mkdir -p /root/repo/tools/experiments/ablate
OBJS=$(ls ../lib/obj/*.o | grep -v psacx_u64.o)
for a in "$@"; do
  k=${a%%=*}; f=${a#*=}
  /opt/rocm/bin/hipcc -O3 -std=c++17 --offload-arch=gfx950 -fPIC -Wno-unused-function -Wno-unused-result $f -c psacx_u64.hip -o /tmp/u64_$k.o &&
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -o /root/repo/tools/experiments/ablate/libpsacx_$k.so /tmp/u64_$k.o $OBJS -ldl &
done
wait
ls -la /root/repo/tools/experiments/ablate
