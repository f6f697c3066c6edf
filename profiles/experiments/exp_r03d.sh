#!/bin/bash
# where does an fp32 launch's time go: ablations (results wrong by construction; MISMATCH expected), same box
O=gpurun_out/r03d; mkdir -p $O
A=tools/bin/fgbench_abl
for cfg in "8192,200,64,12,1,2,1" "8192,200,128,8,1,2,1" "4096,200,64,8,1,0,0"; do
  for v in 0 256 512 1024 1536 2048 4096 3584 3840; do
    timeout -k 10 60 $A reps=40 nt=1 xcd=1 variant=$v $cfg 2>/dev/null | tail -1 | sed "s/^/| variant $v /" >> $O/ablate.md || exit 1
  done
done
cat $O/ablate.md | cut -d'|' -f2-8,11-14
