#!/bin/bash
# round-2 experiment S: deal only part of the tiles XCD-contiguously, the launch's tail in id order (all XCDs share it)
mkdir -p gpurun_out/r02s
O=gpurun_out/r02s
timeout -k 10 300 tools/bin/fgbench reps=40 nt=1 xcd=1 \
  xcdpct=100 4096,200,64,8,1 xcdpct=95 4096,200,64,8,1 xcdpct=90 4096,200,64,8,1 xcdpct=85 4096,200,64,8,1 xcdpct=75 4096,200,64,8,1 xcdpct=50 4096,200,64,8,1 xcdpct=0 4096,200,64,8,1 \
  xcdpct=100 4096,200,64,8,1 xcdpct=90 4096,200,64,8,1 xcdpct=85 4096,200,64,8,1 \
  xcdpct=100 400,2000,64,8,1 xcdpct=90 400,2000,64,8,1 xcdpct=80 400,2000,64,8,1 \
  xcdpct=100 2048,200,64,8,1 xcdpct=85 2048,200,64,8,1 xcdpct=70 2048,200,64,8,1 \
  xcdpct=100 4096,200,64,12,1,0,1 xcdpct=90 4096,200,64,12,1,0,1 xcdpct=80 4096,200,64,12,1,0,1 \
  nt=0 xcdpct=100 1024,200,64,0,1 xcdpct=70 1024,200,64,0,1 xcdpct=50 1024,200,64,0,1 xcdpct=0 1024,200,64,0,1 \
  > $O/fgbench.md 2>&1; echo "fgbench exit $?"; cat $O/fgbench.md
