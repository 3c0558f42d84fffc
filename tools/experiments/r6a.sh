#!/bin/bash
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r6a
./tools/ubench_gather 32 30 > gpurun_out/r6a/ubench_gather_32_30.txt 2>&1
./tools/ubench_gather 27 27 > gpurun_out/r6a/ubench_gather_27_27.txt 2>&1
./tools/ubench_gather 30 30 > gpurun_out/r6a/ubench_gather_30_30.txt 2>&1
python tools/ab_side.py 2 27 1024 3 check > gpurun_out/r6a/twin.txt 2>&1
python tools/ab_side.py 3 30 65536 3 check > gpurun_out/r6a/mutated.txt 2>&1
tools/prof_kind.sh r6a/twin_trace 2 27 1024 1
tools/prof_kind.sh r6a/mut_trace 3 30 65536 1
cat gpurun_out/r6a/*.txt
