#!/bin/bash
# stream-table loads issued before the x window (one memory latency less per wave): previous build vs this one, alternating processes
O=gpurun_out/r03r; mkdir -p $O
{
for rep in 1 2; do
for exe in fgbench_prev fgbench; do
  timeout -k 10 200 tools/bin/$exe reps=80 nt=0 xcd=1 stagger=1 64,200,64,0,1,0,0 128,200,64,0,1,0,0 512,200,64,0,1,0,0 1024,200,64,0,1,0,0 1024,200,64,0,1,2,0 1024,200,128,0,1,2,1 2048,200,64,0,1,2,1 \
     stagger=0 nt=1 4096,200,64,8,1,0,0 8192,200,64,8,1,2,0 8192,200,128,8,1,2,1 pat=1 4096,200,64,8,0,0,0 | tail -11 | cut -d'|' -f2,4,5,6,7,11,13 | sed "s/^/| $exe /" || exit 1
done
done
} > $O/ab.md 2>&1
cat $O/ab.md
timeout -k 10 900 python -m pytest tests -m gpu -x -q > $O/pytest_gpu.log 2>&1; echo "pytest exit $?"; tail -3 $O/pytest_gpu.log
