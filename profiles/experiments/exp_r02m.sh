#!/bin/bash
# round-2 experiment M: shifted slab stream -- GPU suite, fp32 / G7 / odd-ts timings
mkdir -p gpurun_out/r02m
O=gpurun_out/r02m
timeout -k 10 600 python -m pytest tests -m gpu -x -q > $O/pytest_gpu.log 2>&1; echo "pytest exit $?"; tail -5 $O/pytest_gpu.log | cut -c1-400
timeout -k 10 300 tools/bin/fgbench reps=40 nt=1 xcd=1 \
  4096,200,64,8,1 4096,200,64,8,1,1 4096,200,64,12,1,0,1 4096,200,64,12,1,1,1 8192,200,64,12,1,2,1 8192,200,64,8,1,2 4096,201,64,8,1 4096,201,64,12,1,1,1 \
  > $O/fgbench.md 2>&1; echo "fgbench exit $?"; cat $O/fgbench.md
