#!/bin/bash
# how much would full 64-node tiles buy at ts = 200?  ts = 192 and 256 have them (3 / 4 tiles of 64), ts = 200 has 4 x 52
O=gpurun_out/r02an; mkdir -p $O
timeout -k 10 600 tools/bin/fgbench reps=60 nt=1 xcd=1 \
  4096,200,64,8,1 4266,192,64,8,1 3200,256,64,8,1 4096,200,64,8,1 4266,192,64,8,1 3200,256,64,8,1 \
  4096,200,64,12,1,0,1 4266,192,64,12,1,0,1 3200,256,64,12,1,0,1 \
  pat=1 4096,200,64,8,0 4266,192,64,8,0 3200,256,64,8,0 4096,200,64,0,0,0,1 4266,192,64,0,0,0,1 3200,256,64,0,0,0,1 \
  > $O/fgbench.md 2>&1
echo "fgbench exit $?"; cat $O/fgbench.md
