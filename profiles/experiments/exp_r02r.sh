#!/bin/bash
# round-2 experiment R: persistent workgroups with per-XCD tile queues vs one workgroup per tile
mkdir -p gpurun_out/r02r
O=gpurun_out/r02r
timeout -k 10 300 tools/bin/fgbench reps=40 nt=1 xcd=1 \
  persist=0 4096,200,64,8,1 persist=8 4096,200,64,8,1 persist=7 4096,200,64,7,1 persist=6 4096,200,64,6,1 persist=5 4096,200,64,5,1 persist=10 4096,200,64,10,1 persist=6 4096,200,64,8,1 \
  persist=0 4096,200,64,8,1 persist=6 4096,200,64,6,1 persist=7 4096,200,64,7,1 \
  persist=0 400,2000,64,8,1 persist=6 400,2000,64,6,1 persist=7 400,2000,64,7,1 \
  persist=0 8192,200,64,8,1,2 persist=6 8192,200,64,6,1,2 persist=7 8192,200,64,7,1,2 \
  persist=0 4096,200,64,12,1,0,1 persist=8 4096,200,64,8,1,0,1 persist=10 4096,200,64,10,1,0,1 persist=12 4096,200,64,12,1,0,1 \
  persist=0 2048,200,64,8,1 persist=6 2048,200,64,6,1 persist=7 2048,200,64,7,1 \
  > $O/fgbench.md 2>&1; echo "fgbench exit $?"; cat $O/fgbench.md
