#!/bin/bash
# staggered start of the first machine-fill (TOLFG_STAGGER x 0.26 us spread): does breaking the generations' lockstep help?
O=gpurun_out/r02am; mkdir -p $O
A="reps=60 nt=1 xcd=1 4096,200,64,8,1 400,2000,64,8,1 2048,200,64,8,1 nt=0 1024,200,64,0,1"
{
for b in fgbench fgbench_stag8 fgbench_stag24 fgbench_stag48 fgbench; do echo "== $b"; timeout -k 10 200 tools/bin/$b $A | grep -v "^|---\|^| B "; done
} > $O/fgbench.md 2>&1
cat $O/fgbench.md
