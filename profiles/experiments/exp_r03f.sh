#!/bin/bash
# pipelined stream loop (vs exp_r03e on another box: compare within this call only) and issue-priority stagger for launches within the cache
O=gpurun_out/r03f; mkdir -p $O
F=tools/bin/fgbench
{
for st in 0 1; do
timeout -k 10 200 $F reps=80 nt=0 xcd=1 stagger=$st 64,200,64,0,1,0,0 128,200,64,0,1,0,0 256,200,64,0,1,0,0 512,200,64,0,1,0,0 1024,200,64,0,1,0,0 2048,200,64,0,1,0,0 1024,200,64,0,1,2,0 \
   128,200,64,0,1,2,1 1024,200,64,0,1,2,1 1024,200,128,0,1,2,1 2048,200,64,0,1,2,1 2048,200,128,0,1,2,1 50,2000,64,0,1,0,0 \
   nt=1 4096,200,64,8,1,0,0 8192,200,64,12,1,2,1 8192,200,128,8,1,2,1 | tail -16 | sed "s/^/| stagger=$st /" || exit 1
done
} > $O/stagger.md 2>&1
cut -d'|' -f2,3,4,5,6,7,8,9,12,14,15 $O/stagger.md
timeout -k 10 900 python -m pytest tests -m gpu -x -q > $O/pytest_gpu.log 2>&1; echo "pytest exit $?"; tail -5 $O/pytest_gpu.log
