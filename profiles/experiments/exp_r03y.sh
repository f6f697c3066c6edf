#!/bin/bash
# dt = x[0] through a scalar load (variant 0) vs the vector load of rounds 1-2 (variant 65536): same buffers, alternating
O=gpurun_out/r03y; mkdir -p $O
A=tools/bin/fgbench_abl
{
for shape in "nt=0 128,200,64,0,1,0,0" "nt=0 1024,200,64,0,1,0,0" "nt=0 1024,200,128,0,1,2,1" "nt=1 4096,200,64,8,1,0,0" "nt=1 8192,200,64,8,1,2,0" "nt=1 8192,200,128,8,1,2,1" "nt=1 4096,200,64,12,1,0,1" "nt=1 pat=1 4096,200,64,8,0,0,0"; do
  set -- $shape
  last=${@: -1}; opts=${@:1:$#-1}
  timeout -k 10 200 $A reps=100 xcd=1 $opts variant=0 $last variant=65536 $last variant=0 $last variant=65536 $last variant=0 $last variant=65536 $last 2>/dev/null | tail -6 | cut -d'|' -f2,3,4,5,11,14 | tr '\n' ' ' || exit 1
  echo
done
} > $O/dt.md 2>&1
cat $O/dt.md
