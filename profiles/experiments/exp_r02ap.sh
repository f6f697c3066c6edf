#!/bin/bash
# cache-policy bits of the slab stores inside the fg kernel (TOLFG_STORE_FLAVOR builds): 1 nt, 2 sc1, 3 sc0 sc1, 4 sc1 nt, 5 sc0 sc1 nt
O=gpurun_out/r02ap; mkdir -p $O
A="reps=50 nt=1 xcd=1 4096,200,64,8,1 400,2000,64,8,1"
{
for b in fgbench fgbench_fl1 fgbench_fl2 fgbench_fl3 fgbench_fl4 fgbench_fl5 fgbench; do echo "== $b"; timeout -k 10 200 tools/bin/$b $A | grep -v "^|---\|^| B "; done
} > $O/fgbench.md 2>&1
cat $O/fgbench.md
