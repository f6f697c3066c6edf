#!/bin/bash
# finer tiles for the trajectories a launch reaches last (FgArgs::tail_count): same-box A/B with tools/fgbench
O=gpurun_out/r02y; mkdir -p $O
S=4096,200,64,8,1
timeout -k 10 500 tools/bin/fgbench reps=60 nt=1 xcd=1 \
  tail=0 $S tail=128:16 $S tail=256:16 $S tail=512:16 $S tail=1024:16 $S \
  tail=0 $S tail=128:32 $S tail=256:32 $S tail=512:32 $S tail=1024:32 $S \
  tail=0 $S tail=256:8 $S tail=256:24 $S tail=512:24 $S tail=4096:32 $S \
  tail=0 400,2000,64,8,1 tail=16:16 400,2000,64,8,1 tail=32:32 400,2000,64,8,1 tail=64:32 400,2000,64,8,1 \
  tail=0 4096,200,64,12,1,0,1 tail=256:16 4096,200,64,12,1,0,1 tail=512:32 4096,200,64,12,1,0,1 \
  nt=0 tail=0 1024,200,64,0,1 tail=128:16 1024,200,64,0,1 tail=256:32 1024,200,64,0,1 tail=128:32 1024,200,64,0,1 \
  > $O/fgbench.md 2>&1
echo "fgbench exit $?"; cat $O/fgbench.md
