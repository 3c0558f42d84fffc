#!/bin/bash
# Builds variants of libpsacx.so with other compile-time settings of the ANSV kernel (ansv_wave.hpp), e.g. parts of the furthest_eq pass
# left out to time them (AW_ABLATE bits: 1 pointer doubling, 2 links, 4 the answers out, 8 the carried table; results are wrong then).
# usage: ansv_ablate.sh tag=-DAW_ABLATE=7 tag2="-DAW_FINAL_W=8" ...   Run here; the variants travel with the snapshot (PSACX_LIB selects one).
cd /root/repo/psac_amd/csrc
mkdir -p /root/repo/tools/experiments/ablate
OBJS=$(ls ../lib/obj/*.o | grep -v ansv.o)
for a in "$@"; do
  k=${a%%=*}; f=${a#*=}
  /opt/rocm/bin/hipcc -O3 -std=c++17 --offload-arch=gfx950 -fPIC -Wno-unused-function -Wno-unused-result $f -c ansv.hip -o /tmp/ansv_$k.o &&
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -o /root/repo/tools/experiments/ablate/libpsacx_$k.so /tmp/ansv_$k.o $OBJS -ldl &
done
wait
ls -la /root/repo/tools/experiments/ablate
