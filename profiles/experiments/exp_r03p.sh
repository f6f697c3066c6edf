#!/bin/bash
# stagger on/off alternating on the SAME buffers (one process per shape), three fresh processes each: is the gain real or allocation luck?
O=gpurun_out/r03p; mkdir -p $O
F=tools/bin/fgbench
{
for rep in 1 2 3; do
for shape in 1024,200,64,0,1,0,0 1024,200,64,0,1,2,0 1024,200,128,0,1,2,1 2048,200,64,0,1,2,1 1536,200,64,0,1,0,0; do
timeout -k 10 100 $F reps=100 nt=0 xcd=1 stagger=0 $shape stagger=1 $shape stagger=0 $shape stagger=1 $shape stagger=0 $shape stagger=1 $shape | tail -6 | cut -d'|' -f2,4,5,6,11 | tr '\n' ' ' || exit 1
echo
done
done
} > $O/alt.md 2>&1
cat $O/alt.md
