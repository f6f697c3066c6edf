#!/bin/bash
# compact pattern: where does a launch's time go (ablations; results wrong by construction), fp64 and fp32, same box
O=gpurun_out/r03l; mkdir -p $O
A=tools/bin/fgbench_abl
for cfg in "4096,200,64,8,0,0,0" "4096,200,64,0,0,0,1"; do
  for v in 0 256 512 1024 1536 2048 4096 3584 3840; do
    timeout -k 10 60 $A reps=40 nt=1 xcd=1 pat=1 variant=$v $cfg 2>/dev/null | tail -1 | sed "s/^/| variant $v /" >> $O/ablate.md || exit 1
  done
done
cut -d'|' -f2-8,11-14 $O/ablate.md
