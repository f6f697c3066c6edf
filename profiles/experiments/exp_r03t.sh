#!/bin/bash
# table loads before (variant 0) vs after (variant 8192) the x window: the SAME buffers, alternating, ablation build (both orders compiled in)
O=gpurun_out/r03t; mkdir -p $O
A=tools/bin/fgbench_abl
{
for shape in "nt=0 128,200,64,0,1,0,0" "nt=0 1024,200,64,0,1,0,0" "nt=0 1024,200,128,0,1,2,1" "nt=1 4096,200,64,8,1,0,0" "nt=1 8192,200,64,8,1,2,0" "nt=1 8192,200,128,8,1,2,1" "nt=1 8192,200,64,12,1,2,1" "nt=1 pat=1 4096,200,64,8,0,0,0" "nt=1 pat=1 4096,200,64,0,0,0,1"; do
  set -- $shape
  last=${@: -1}; opts=${@:1:$#-1}
  timeout -k 10 200 $A reps=60 xcd=1 $opts variant=0 $last variant=8192 $last variant=0 $last variant=8192 $last variant=0 $last variant=8192 $last 2>/dev/null | tail -6 | cut -d'|' -f2,4,5,6,11 | tr '\n' ' ' || exit 1
  echo
done
} > $O/order.md 2>&1
cat $O/order.md
