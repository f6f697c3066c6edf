#!/bin/bash
# finalize reads its 23 edge values at the wave's start (variant 0) vs at the end (variant 16384): same buffers, alternating; then the GPU suite
O=gpurun_out/r03w; mkdir -p $O
A=tools/bin/fgbench_abl
{
for shape in "nt=0 8,2000,64,0,1,0,0" "nt=0 64,200,64,0,1,0,0" "nt=0 128,200,64,0,1,0,0" "nt=0 256,200,64,0,1,0,0" "nt=0 1024,200,64,0,1,0,0" "nt=0 1024,200,128,0,1,2,1" "nt=1 4096,200,64,8,1,0,0" "nt=1 8192,200,64,8,1,2,0" "nt=1 8192,200,128,8,1,2,1"; do
  set -- $shape
  last=${@: -1}; opts=${@:1:$#-1}
  timeout -k 10 200 $A reps=100 xcd=1 $opts variant=0 $last variant=16384 $last variant=0 $last variant=16384 $last variant=0 $last variant=16384 $last 2>/dev/null | tail -6 | cut -d'|' -f2,3,4,5,11 | tr '\n' ' ' || exit 1
  echo
done
} > $O/edge.md 2>&1
cat $O/edge.md
timeout -k 10 900 python -m pytest tests -m gpu -x -q > $O/pytest_gpu.log 2>&1; echo "pytest exit $?"; tail -3 $O/pytest_gpu.log
