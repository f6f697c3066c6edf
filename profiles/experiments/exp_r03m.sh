#!/bin/bash
# x-window prefetch by leaving waves (FgArgs::prefetch = tiles ahead on the XCD's run): distance sweep, same box
O=gpurun_out/r03m; mkdir -p $O
F=tools/bin/fgbench
{
for pf in 0 128 256 320 384 512 768 0; do
  timeout -k 10 120 $F reps=60 nt=1 xcd=1 prefetch=$pf pat=1 4096,200,64,8,0,0,0 4096,200,64,0,0,0,1 pat=0 4096,200,64,8,1,0,0 8192,200,64,8,1,2,0 8192,200,64,12,1,2,1 8192,200,128,8,1,2,1 | tail -6 | sed "s/^/| pf=$pf /" || exit 1
done
} > $O/prefetch.md 2>&1
cut -d'|' -f2,3,4,5,6,7,8,9,12,14,15 $O/prefetch.md
