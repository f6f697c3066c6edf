#!/bin/bash
# compact pattern, fp64: resident-wave cap, uninstrumented time column
O=gpurun_out/r02al; mkdir -p $O
timeout -k 10 600 tools/bin/fgbench reps=60 pat=1 nt=1 xcd=1 \
  4096,200,64,8,0 4096,200,64,0,0 4096,200,64,12,0 4096,200,64,10,0 4096,200,64,8,0 4096,200,64,0,0 \
  400,2000,64,8,0 400,2000,64,0,0 8192,200,64,8,0,2 8192,200,64,0,0,2 4096,200,64,8,0,1 4096,200,64,0,0,1 \
  > $O/fgbench.md 2>&1
echo "fgbench exit $?"; cat $O/fgbench.md
