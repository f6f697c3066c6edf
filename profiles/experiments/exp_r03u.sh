#!/bin/bash
# slab stream: 1 / 2 / 3 rounds of LDS reads in flight ahead of the store (separate binaries: compare the small-B rows, where one wave
# is alone on its SIMD); and a pure store loop at low occupancy (wrbench) for the per-wave store rate
O=gpurun_out/r03u; mkdir -p $O
{
for rep in 1 2; do
for d in 1 2 3; do
  timeout -k 10 120 tools/bin/fgbench_a$d reps=200 nt=0 xcd=1 1,200,64,0,1,0,0 64,200,64,0,1,0,0 128,200,64,0,1,0,0 256,200,64,0,1,0,0 1024,200,64,0,1,0,0 nt=1 4096,200,64,8,1,0,0 | tail -6 | cut -d'|' -f2,4,5,11 | tr '\n' ' ' | sed "s/^/ahead=$d /" || exit 1
  echo
done
done
for lds in 65536 40960 20480 0; do timeout -k 5 60 tools/bin/wrbench 4 43 $lds; timeout -k 5 60 tools/bin/wrbench 0 43 $lds; done
} > $O/ahead.md 2>&1
cat $O/ahead.md
