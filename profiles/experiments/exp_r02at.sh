#!/bin/bash
# which 16-byte position of a row's slab region inside a 128-byte line is fast?  ldgpad=2: every row the same position; goff shifts it
O=gpurun_out/r02at; mkdir -p $O
S=4096,200,64,8,1
timeout -k 10 800 tools/bin/fgbench reps=50 nt=1 xcd=1 \
  ldgpad=0 goff=0 $S \
  ldgpad=2 goff=0 $S goff=2 $S goff=4 $S goff=6 $S goff=8 $S goff=10 $S goff=12 $S goff=14 $S \
  ldgpad=0 goff=0 $S \
  ldgpad=2 goff=0 $S goff=2 $S goff=4 $S goff=6 $S goff=8 $S goff=10 $S goff=12 $S goff=14 $S \
  > $O/fgbench.md 2>&1
echo "fgbench exit $?"; cat $O/fgbench.md | cut -d'|' -f11-13
