#!/bin/bash
# round-2 experiment V: compact pattern -- persistent form and caps (fused / unfused)
mkdir -p gpurun_out/r02v
O=gpurun_out/r02v
timeout -k 10 300 tools/bin/fgbench_persist reps=40 pat=1 nt=1 xcd=1 \
  persist=0 4096,200,64,8,0 4096,200,64,0,0 4096,200,64,8,1 \
  persist=8 4096,200,64,8,1 persist=10 4096,200,64,10,1 persist=6 4096,200,64,6,1 persist=12 4096,200,64,0,1 \
  persist=0 4096,200,64,12,0,0,1 persist=12 4096,200,64,12,1,0,1 persist=16 4096,200,64,16,1,0,1 \
  > $O/fgbench.md 2>&1; echo "exit $?"; cat $O/fgbench.md
