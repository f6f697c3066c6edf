#!/bin/bash
# finer-tiled tail at the batch sizes whose outputs fit the Infinity Cache (plain stores, no cap) and at B=2048
O=gpurun_out/r02z; mkdir -p $O
timeout -k 10 500 tools/bin/fgbench reps=80 nt=0 xcd=1 \
  tail=0 1024,200,64,0,1 tail=256:32 1024,200,64,0,1 tail=384:32 1024,200,64,0,1 tail=512:32 1024,200,64,0,1 tail=1024:32 1024,200,64,0,1 \
  tail=256:40 1024,200,64,0,1 tail=512:40 1024,200,64,0,1 tail=1024:40 1024,200,64,0,1 tail=0 1024,200,64,0,1 \
  tail=0 512,200,64,0,1 tail=128:32 512,200,64,0,1 tail=256:32 512,200,64,0,1 tail=512:32 512,200,64,0,1 tail=256:16 512,200,64,0,1 \
  tail=0 128,200,64,0,1 tail=64:32 128,200,64,0,1 tail=128:32 128,200,64,0,1 tail=128:16 128,200,64,0,1 \
  tail=0 256,200,64,0,1 tail=128:32 256,200,64,0,1 tail=256:32 256,200,64,0,1 \
  nt=1 tail=0 2048,200,64,8,1 tail=128:32 2048,200,64,8,1 tail=256:32 2048,200,64,8,1 \
  nt=0 tail=0 2048,200,64,0,1 tail=256:32 2048,200,64,0,1 \
  nt=0 tail=0 1024,200,64,0,1,2 tail=256:32 1024,200,64,0,1,2 tail=0 1024,200,64,0,1,0,1 tail=256:32 1024,200,64,0,1,0,1 \
  > $O/fgbench.md 2>&1
echo "fgbench exit $?"; cat $O/fgbench.md
