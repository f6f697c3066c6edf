#!/bin/bash
# tile size vs batch size for launches that fit the cache (plain stores, no cap): what should plan_launch choose?
O=gpurun_out/r03e; mkdir -p $O
F=tools/bin/fgbench
{
for B in 64 128 256 512 1024 2048; do
  timeout -k 10 120 $F reps=80 nt=0 xcd=1 $B,200,64,0,1,0,0 $B,200,52,0,1,0,0 $B,200,40,0,1,0,0 $B,200,36,0,1,0,0 $B,200,28,0,1,0,0 $B,200,20,0,1,0,0 $B,200,16,0,1,0,0 $B,200,12,0,1,0,0 $B,200,8,0,1,0,0 | tail -9 || exit 1
done
for B in 128 256 1024 2048; do
  timeout -k 10 120 $F reps=80 nt=0 xcd=1 $B,200,128,0,1,2,1 $B,200,64,0,1,2,1 $B,200,40,0,1,2,1 $B,200,28,0,1,2,1 $B,200,20,0,1,2,1 $B,200,16,0,1,2,1 $B,200,8,0,1,2,1 | tail -7 || exit 1
done
timeout -k 10 120 $F reps=80 nt=0 xcd=1 1024,200,64,0,1,2,0 1024,200,40,0,1,2,0 1024,200,28,0,1,2,0 50,2000,64,0,1,0,0 50,2000,32,0,1,0,0 50,2000,16,0,1,0,0 | tail -6
} > $O/tiles.md 2>&1
cut -d'|' -f2,3,4,5,6,8,11,13,14 $O/tiles.md
