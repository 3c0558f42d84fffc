#!/bin/bash
# kept evidence for the numbers DESIGN / README quote
cd $GRAFT_REPO_ROOT
export PSACX_ENV_KNOBS=1
O=gpurun_out/r6n; mkdir -p $O
bash tools/ansv_pmc.sh r6n/ansv_t t > $O/ansv_t.log 2>&1
PSACX_MULTI_TRACE=1 timeout 600 python tools/dist_bigrun.py 1 32 64 tandem reduced > $O/multi_tandem_1x2p32.txt 2>&1
PSACX_MULTI_TRACE=1 timeout 600 python tools/dist_bigrun.py 8 28 64 tandem reduced > $O/multi_tandem_8x2p28.txt 2>&1
PSACX_MULTI_TRACE=1 PSACX_MULTI_FORCE_WIRE=1 timeout 600 python tools/dist_bigrun.py 1 31 64 dna reduced > $O/multi_dna_1x2p31_wire.txt 2>&1
PSACX_MULTI_TRACE=1 timeout 600 python tools/dist_bigrun.py 8 28 64 dna reduced > $O/multi_dna_8x2p28.txt 2>&1
cd /tmp; export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
for w in "mutated 3 30 65536" "tandem 2 30 1024"; do
  set -- $w
  for ctr in FETCH_SIZE WRITE_SIZE; do
    rocprofv3 --pmc $ctr -d /tmp/r6n/$1_$ctr -o a -- python $R/tools/ab_side.py $2 $3 $4 1 > $R/$O/$1_$ctr.log 2>&1
    python3 $R/tools/rocpd_summary.py /tmp/r6n/$1_$ctr/a_results.db > $R/$O/refine_${ctr}_$1_2p30.txt 2>&1
    rm -rf /tmp/r6n/$1_$ctr
  done
done
cd $R; tail -3 $O/multi_*.txt | cut -c1-300
