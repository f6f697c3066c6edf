#!/bin/bash
# do 128-byte-aligned slab regions matter?  ldgpad=2 makes the row stride a multiple of 128 B; goff=4 then puts every row's slab
# region (row + c0) on a 128-byte line, so that no line is shared between two tile waves
O=gpurun_out/r02as; mkdir -p $O
S=4096,200,64,8,1
timeout -k 10 700 tools/bin/fgbench reps=50 nt=1 xcd=1 \
  ldgpad=0 goff=0 $S ldgpad=2 goff=0 $S ldgpad=2 goff=4 $S ldgpad=2 goff=12 $S ldgpad=0 goff=0 $S ldgpad=2 goff=4 $S ldgpad=2 goff=0 $S ldgpad=2 goff=4 $S \
  > $O/fgbench.md 2>&1
echo "fgbench exit $?"; cat $O/fgbench.md
