#!/bin/bash
# s_setprio experiments: 1 = store phase at priority 3, 2 = load/compute phase at priority 3 (dropped to 0 for the stores)
O=gpurun_out/r02ac; mkdir -p $O
A="reps=60 nt=1 xcd=1 4096,200,64,8,1 400,2000,64,8,1 4096,200,64,12,1,0,1 8192,200,64,8,1,2 nt=0 1024,200,64,0,1"
{
for b in fgbench fgbench_prio1 fgbench_prio2 fgbench; do echo "== $b"; timeout -k 10 200 tools/bin/$b $A; done
} > $O/fgbench.md 2>&1
echo "exit $?"; grep -v "^|---\|^| B " $O/fgbench.md
