#!/bin/bash
# launches within the cache: resident-wave cap x issue-priority stagger (B = 1024 / 2048, fp64 and fp32), same box
O=gpurun_out/r03o; mkdir -p $O
F=tools/bin/fgbench
{
for st in 0 1; do
timeout -k 10 200 $F reps=100 nt=0 xcd=1 stagger=$st 1024,200,64,0,1,0,0 1024,200,64,8,1,0,0 1024,200,64,6,1,0,0 1024,200,64,5,1,0,0 1024,200,64,4,1,0,0 1024,200,64,0,1,0,0 \
   1024,200,64,0,1,2,0 1024,200,64,8,1,2,0 1024,200,64,0,1,2,1 1024,200,64,8,1,2,1 1024,200,64,12,1,2,1 1024,200,128,0,1,2,1 1024,200,128,4,1,2,1 \
   2048,200,64,0,1,0,0 2048,200,64,8,1,0,0 | tail -15 | sed "s/^/| stagger=$st /" || exit 1
done
} > $O/cap_stagger.md 2>&1
cut -d'|' -f2,3,4,5,6,7,8,9,12,14,15 $O/cap_stagger.md
