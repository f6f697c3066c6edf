#!/bin/bash
# does the row stride of G (the spacing of the concurrent store fronts) matter?  ldgpad = extra elements between rows
O=gpurun_out/r02ao; mkdir -p $O
S=4096,200,64,8,1
timeout -k 10 700 tools/bin/fgbench reps=50 nt=1 xcd=1 \
  ldgpad=0 $S ldgpad=2 $S ldgpad=10 $S ldgpad=66 $S ldgpad=74 $S ldgpad=514 $S ldgpad=1090 $S ldgpad=2050 $S ldgpad=3138 $S ldgpad=11330 $S ldgpad=0 $S \
  > $O/fgbench.md 2>&1
echo "fgbench exit $?"; cat $O/fgbench.md
