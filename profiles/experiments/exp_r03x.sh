#!/bin/bash
# what do the per-wave loads of the stream table cost?  ablation: offsets made up in registers (variant 32768, results wrong) vs loaded (0); same buffers
O=gpurun_out/r03x; mkdir -p $O
A=tools/bin/fgbench_abl
{
for shape in "nt=0 128,200,64,0,1,0,0" "nt=0 1024,200,64,0,1,0,0" "nt=1 4096,200,64,8,1,0,0" "nt=1 8192,200,64,8,1,2,0" "nt=1 8192,200,128,8,1,2,1" "nt=1 8192,200,64,12,1,2,1" "nt=1 pat=1 4096,200,64,8,0,0,0"; do
  set -- $shape
  last=${@: -1}; opts=${@:1:$#-1}
  timeout -k 10 200 $A reps=100 xcd=1 $opts variant=0 $last variant=32768 $last variant=0 $last variant=32768 $last 2>/dev/null | tail -4 | cut -d'|' -f2,3,4,5,11 | tr '\n' ' ' || exit 1
  echo
done
} > $O/table_cost.md 2>&1
cat $O/table_cost.md
