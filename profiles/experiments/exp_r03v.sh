#!/bin/bash
# compiler scheduling strategy: default (max occupancy) vs -mllvm -amdgpu-sched-strategy=max-ilp (separate binaries, two passes)
O=gpurun_out/r03v; mkdir -p $O
{
for rep in 1 2; do
for exe in fgbench fgbench_ilp; do
  timeout -k 10 120 tools/bin/$exe reps=200 nt=0 xcd=1 1,200,64,0,1,0,0 64,200,64,0,1,0,0 128,200,64,0,1,0,0 512,200,64,0,1,0,0 1024,200,64,0,1,0,0 1024,200,128,0,1,2,1 nt=1 4096,200,64,8,1,0,0 8192,200,128,8,1,2,1 | tail -8 | cut -d'|' -f2,4,5,11 | tr '\n' ' ' | sed "s/^/$exe /" || exit 1
  echo
done
done
} > $O/ilp.md 2>&1
cat $O/ilp.md
